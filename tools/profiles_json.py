"""Turns the text summaries of tools/pmc.sh into the two small JSON files bench.py reads:
   traffic*.json       HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (KiB; the gfx950 correction of MI355X_MICROARCH.md)
   executed_fp64.json  FP64 operations the kernel EXECUTED per launch = 64 x (ADD + MUL + 2 FMA + TRANS) wavefront instructions
usage: profiles_json.py <profiles/rNN> summary.txt:variant:H:N[:kernel-substring[:dispatches per launch]] ..."""
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


def main():
    dst = sys.argv[1]
    fp64, traffic = [], []
    for spec in sys.argv[2:]:
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
