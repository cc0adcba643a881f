"""Turns the text summaries of tools/pmc.sh into the two small JSON files bench.py reads:
   traffic*.json       HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (KiB; the gfx950 correction of MI355X_MICROARCH.md)
   executed_fp64.json  FP64 operations the kernel EXECUTED per launch = 64 x (ADD + MUL + 2 FMA + TRANS) wavefront instructions
   qp_roofline.json    (--qp summary.txt:H:QPs per launch:kernel_stats.csv) the same two figures per QP step of the solver's
                       dominant kernel, its lane occupancy, and its share of the GPU time of a full-size solve
usage: profiles_json.py <profiles/rNN> [--qp pmc_summary_qp.txt:H:K:kernel_stats_solve.csv[:kernel_trace_solve.csv]]
                        (with the trace: the share is the time some QP kernel was running / the span of the last solve in it,
                         from its solve_begin_kernel to its last kernel's end - a round's evaluation kernels run beside its QP
                         kernel on a second stream and wait there for room, so their summed durations say nothing)
                        summary.txt:variant:H:N[:kernel-substring[:dispatches per launch]] ..."""
import json, os, re, sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("== "):
            cur = out.setdefault(line[3:].strip(), {})
        else:
            m = re.match(r"(\S+)\s+n=(\d+) mean=(\S+)", line)
            if m and cur is not None:
                cur[m.group(1)] = float(m.group(3))
    return out


def qp_roofline(dst, spec):
    """pmc_qp.sh's flat summary (one kernel) + the kernel trace of a full-size resident solve -> qp_roofline.json"""
    import csv
    parts = spec.split(":")
    path, H, K, stats = parts[:4]
    trace = parts[4] if len(parts) > 4 else None
    c = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+n=(\d+) mean=(\S+)", line)
        if m:
            c[m.group(1)] = float(m.group(3))
    K = float(K)
    flops = 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"])
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    qp = [r for r in rows if "qp_step_kernel" in r["Name"]]
    occ = (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
           if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU") else None)
    out = {"kernel": "qp_step_kernel<true, true, false>", "horizon": int(H), "qps_per_counter_launch": K,
           "wave_instructions_per_launch": {k[14:]: c[k] for k in c if k.startswith("SQ_INSTS_VALU_") and k.endswith("F64")},
           "executed_flops_per_qp": flops / K, "lane_occupancy": occ,
           "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"], "fetch_correction": 2.0,
           "traffic_bytes_per_qp": 1024.0 * (c["WRITE_SIZE"] + 2.0 * c["FETCH_SIZE"]) / K,
           "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else None,
           "wait_share_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in c else None,
           "share_of_gpu_time": sum(float(r["TotalDurationNs"]) for r in qp) / total,
           "share_source": os.path.basename(stats), "source": os.path.basename(path),
           "note": "counters over K recorded QP subproblems of real solves (tools/qp_time.py); the share is the kernel's part of "
                   "the summed kernel time of one 65,536-scenario resident solve (rocprofv3 --kernel-trace --stats)"}
    if trace:
        k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace)))
        i0 = max(i for i, (_, _, n) in enumerate(k) if "solve_begin_kernel" in n)
        s0, e1 = k[i0][0], max(e for _, e, _ in k[i0:])
        busy, cs, ce = 0, None, None
        for s_, e_, n in k[i0:]:
            if "qp_step_kernel" not in n:
                continue
            if ce is None or s_ > ce:
                busy += (ce - cs) if ce is not None else 0
                cs, ce = s_, e_
            else:
                ce = max(ce, e_)
        busy += (ce - cs) if ce is not None else 0
        out.update({"share_of_gpu_time": busy / (e1 - s0), "solve_span_ms": (e1 - s0) / 1e6, "qp_kernel_running_ms": busy / 1e6,
                    "share_source": os.path.basename(trace),
                    "note": "counters over K recorded QP subproblems of real solves (tools/qp_time.py); the share is the time a QP "
                            "kernel was running / the span of one 65,536-scenario resident solve in rocprofv3's kernel trace "
                            "(the evaluation kernels of a round run beside its QP kernel on a second stream)"})
    json.dump(out, open(os.path.join(dst, "qp_roofline.json"), "w"), indent=1)
    print(f"qp_roofline.json: {out['executed_flops_per_qp'] / 1e6:.2f} Mflop and {out['traffic_bytes_per_qp'] / 1e6:.2f} MB per QP, "
          f"share {out['share_of_gpu_time']:.3f}")


def main():
    dst = sys.argv[1]
    fp64, traffic = [], []
    specs = sys.argv[2:]
    if specs and specs[0] == "--qp":
        qp_roofline(dst, specs[1])
        specs = specs[2:]
    for spec in specs:
        parts = spec.split(":")
        path, variant, H, N = parts[0], parts[1], int(parts[2]), int(parts[3])
        want = parts[4] if len(parts) > 4 else "prefix_kernel"
        mult = float(parts[5]) if len(parts) > 5 else 1.0      # a launch that the library cuts into several dispatches
        for kernel, c in parse(path).items():
            if want not in kernel or "SQ_INSTS_VALU_FMA_F64" not in c:
                continue
            flops = mult * 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]
                            + c["SQ_INSTS_VALU_TRANS_F64"])
            fp64.append({"kernel": kernel, "variant": variant, "horizon": H, "scenarios": N,
                         "wave_instructions": {k[14:]: c[k] for k in c if k.startswith("SQ_INSTS_VALU_") and k.endswith("F64")},
                         "valu_instructions": c.get("SQ_INSTS_VALU"), "executed_flops_per_launch": flops,
                         "lane_occupancy": (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
                                            if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU") else None),
                         "source": os.path.relpath(path, os.path.dirname(dst.rstrip("/")) or ".")})
            if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
                traffic.append({"kernel": kernel, "variant": variant, "horizon": H, "scenarios": N, "algo": "prefix",
                                "WRITE_SIZE_KiB": c["WRITE_SIZE"], "FETCH_SIZE_KiB": c["FETCH_SIZE"], "fetch_correction": 2.0,
                                "dispatches_per_launch": mult,
                                "traffic_bytes_per_launch": mult * 1024.0 * (c["WRITE_SIZE"] + 2.0 * c["FETCH_SIZE"]),
                                "source": os.path.basename(path)})
    json.dump(fp64, open(os.path.join(dst, "executed_fp64.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(f"{len(fp64)} fp64 entries, {len(traffic)} traffic entries -> {dst}")


if __name__ == "__main__":
    main()
