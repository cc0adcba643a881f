#!/usr/bin/env python3
"""Benchmark of the MPC inner loop on MI355X: objective + constraint + FD-gradient evals/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scenarios S] [--horizon H]

A "step" = one mpc_eval_fd pass over the batch resident in HBM: for every scenario f, c, the
forward-difference gradient and the dense forward-difference constraint Jacobian (what scipy's
SLSQP requests per major iteration from the reference's callables, ur_mpc.py:188-200).  Default
workload: 65,536 linear-platform-motion scenarios per GPU at H=20 (the size BASELINE.json's
north_star quotes its target on; configs[1]'s 4,096-scenario rate is reported alongside).  For
N > 1 the driver launches one rank per GPU (torch.distributed, backend nccl = RCCL); scenarios are
sharded contiguously, nothing is exchanged per step, one RCCL all_gather of the per-scenario costs
closes the run (weak scaling).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mpc_amd import _ffi  # noqa: E402
from mpc_amd.batch import EvalResult, MpcContext  # noqa: E402
from mpc_amd.params import MpcParams  # noqa: E402
from mpc_amd.scenarios import SWEEP_ROWS, DeviceKinematics, make_scenarios, make_sweep, to_device  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def log(msg):
    """Progress on stderr when MPC_BENCH_VERBOSE is set (stdout carries only the JSON line)."""
    if os.environ.get("MPC_BENCH_VERBOSE"):
        print(f"[bench rank {os.environ.get('RANK', '0')} {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


_HOST_COLLECTIVES = False    # --rehearse-on-one-gpu: gloo, tensors staged through the host


def all_reduce(dist, t, op):
    if _HOST_COLLECTIVES:
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def all_gather(dist, out, t):
    if _HOST_COLLECTIVES:
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, t.cpu())
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, t)


def launch_ranks(gpus: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as CHILD processes
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) before this process has made any GPU
    call, relay rank 0's JSON line, and return the children's exit status.  The parent never initialises the
    GPU and never re-executes itself."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    if "OMP_NUM_THREADS" not in env:          # torch.distributed.run would pin every rank to ONE host thread
        from mpc_amd.pool import host_core_budget

        env["OMP_NUM_THREADS"] = str(host_core_budget(gpus))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = 0
    for line in proc.stdout:
        if line.lstrip().startswith("{") and '"metric"' in line:
            print(line.rstrip("\n"), flush=True)
            lines += 1
        else:                                  # anything else a rank printed goes to stderr: stdout carries one line
            print(line.rstrip("\n"), file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the ranks exited cleanly but printed {lines} result lines", file=sys.stderr)
        return 1
    return rc


def solve_result_rows(control: np.ndarray, last, rows: np.ndarray) -> np.ndarray:
    """Per-scenario results of one batched tick, [count, 9] = [f*, u0(6), nit, status]:
    what compute_control returns (ur_mpc.py:205-237) plus the optimiser's final cost, iteration count and exit
    mode.  Controllers that took the early exit (ur_mpc.py:108) have f* = 0, nit = 0, status = -1."""
    out = np.zeros((control.shape[0], 9))
    out[:, 1:7] = control
    out[:, 8] = -1.0
    if last is not None and len(rows):
        out[rows, 0] = last.fun
        out[rows, 7] = last.nit
        out[rows, 8] = last.status
    return out


def gather_result_rows(dist, local: np.ndarray, world: int, rank: int, device) -> np.ndarray:
    """The one exchange of the job: every rank's [count, 9] rows -> the [world*count, 9] table in rank order on
    every rank (RCCL all_gather over xGMI; gloo through the host in the rehearsal modes).  Checks that this
    rank's block of the table is bit for bit what it sent."""
    if dist is None:
        return local
    from mpc_amd.sharding import gather_rows

    t = torch.as_tensor(np.ascontiguousarray(local))
    if not _HOST_COLLECTIVES:
        t = t.to(device)
    full = gather_rows(t, world * local.shape[0]).cpu().numpy()
    c = local.shape[0]
    if not np.array_equal(full[rank * c:(rank + 1) * c], local):
        raise RuntimeError("gathered solve results differ from what this rank contributed")
    return full


def algorithmic_flops(H: int) -> float:
    """full rollouts, one fused cost+constraint rollout per perturbation."""
    return (1 + 6 * H) * (745 * H + 600 + 30 * (H + 1))


def algorithmic_bytes(H: int) -> float:
    """inputs read once + dense outputs (cost, grad, c, dc/dx) per eval."""
    n, m = 6 * H, 12 * (H + 1)
    return 8.0 * ((18 * H + 66) + (1 + n + m + m * n))


def cpu_baseline(mp: MpcParams, hs, budget_s: float = 12.0):
    """The reference-shaped numpy/scipy path (oracle/numpy_ref.py, pinned bit-for-bit to the real
    reference) timed on one host core on a bounded sample of the same scenarios; the C port of the
    same path is timed beside it.  The oracle is only the baseline here, never the product."""
    from oracle.numpy_ref import NumpyRefMPC, base_traj_from_array
    from oracle.oracle import COracle, OracleKinematics as _OracleKin, ref_params_from

    rp = ref_params_from(mp)
    t0 = time.perf_counter()
    done = 0
    while done < hs.count and (time.perf_counter() - t0) < budget_s:
        i = done
        ref = NumpyRefMPC(rp)
        ref.J_cached = hs.Jc[i].reshape(6, 6)
        tgt = {"position": hs.target[i, :3], "orientation": hs.target[i, 3:]}
        ref.fd_eval(hs.x[i], hs.state0[i], tgt, base_traj_from_array(hs.base_traj[i]), fast_constraint=False)
        done += 1
    t_np = time.perf_counter() - t0
    co = COracle(rp)
    nC = min(hs.count, 256)
    t1 = time.perf_counter()
    co.eval_fd(hs.as_oracle_dict(0, nC))
    t_c = time.perf_counter() - t1
    # one full solve of the reference-shaped path (scipy minimize over the numpy callables), configs[2] shape
    ha = make_scenarios(mp, _OracleKin(co), 1, motion="angular", cfg_id=3, goal_jitter=0.0)
    ref = NumpyRefMPC(rp)
    info = {}
    t2 = time.perf_counter()
    ref.compute_control({"position": ha.q0[0], "velocity": np.zeros(6)},
                        {"position": ha.state0[0, 12:15], "orientation": ha.state0[0, 15:18]},
                        {"position": ha.target[0, :3], "orientation": ha.target[0, 3:]},
                        {"position": ha.base[0, 0:3], "orientation": ha.base[0, 3:6], "linear_velocity": ha.base[0, 6:9],
                         "angular_velocity": ha.base[0, 9:12]}, info=info)
    t_solve = time.perf_counter() - t2
    from mpc_amd.pool import host_core_budget
    from oracle.cpu_bench import all_cores_evals_per_s

    cores = host_core_budget()
    rows = {k: v for k, v in hs.as_oracle_dict(0, min(hs.count, 4 * cores)).items() if v is not None and k != "has_prev"}
    box_rate, box_wall = all_cores_evals_per_s(rp, rows, cores, budget=8.0)
    return {
        "box_value": box_rate, "box_cores": cores, "box_sample": f"the same numpy path in {cores} processes at once, "
        f"{box_wall:.1f} s",
        "solves_value": 1.0 / t_solve, "solves_unit": "solves/s", "solves_sample": f"1 angular-motion scenario, "
        f"scipy minimize(SLSQP) over the numpy callables, {info.get('nit')} iterations, {info.get('nfev')} cost calls, "
        f"{t_solve:.1f} s, single thread",
        "value": done / t_np, "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": f"{done} scenarios of the same batch, numpy restatement of ur_mpc.py with scipy approx_derivative "
                  f"(1+n cost rollouts + 1+n constraint rollouts per eval), single thread, {t_np:.1f} s",
        "c_port_value": nC / t_c, "c_port_sample": f"{nC} scenarios, scalar C port (oracle/mpc_oracle.c), single thread",
        "host_cpus": os.cpu_count(),
    }


def _profile_entries(name):
    """Entries of profiles/*/<name> (newest round first); a file holds one entry or a list of them."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", name)), reverse=True):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        for e in (t if isinstance(t, list) else [t]):
            yield e, os.path.relpath(path, ROOT)


def measured_traffic(H, S, algo, variant="jac"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md), if a measurement of this exact configuration exists."""
    for t, src in _profile_entries("traffic.json"):
        if (t.get("horizon") == H and t.get("scenarios") == S and t.get("algo") == algo
                and t.get("variant", "jac") == variant):
            return float(t["traffic_bytes_per_launch"]), src
    return None, None


def executed_fp64(H, S, variant, kernel_ms):
    """FP64 operations the kernel EXECUTED, from the committed counter passes of this kernel variant at this horizon
    (64 x (ADD + MUL + 2 FMA + TRANS) wavefront instructions; per scenario the count does not depend on the batch),
    over the live kernel time: what the hardware did, against the vector-FP64 peak.  None without a profile."""
    wanted = (variant, "rollout") if H > 40 else (variant,)      # long horizons: one rollout kernel serves every variant
    for t, src in _profile_entries("executed_fp64.json"):
        if t.get("horizon") == H and t.get("variant") in wanted and "expand" not in t.get("kernel", ""):
            flops = float(t["executed_flops_per_launch"]) / float(t["scenarios"]) * S
            tf = flops / (kernel_ms * 1e-3) / 1e12
            occ = t.get("lane_occupancy")
            out = {"executed_tflops_all_64_lanes": tf, "executed_frac_of_fp64_valu_peak_all_64_lanes": tf / FP64_VALU_PEAK_TF,
                   "executed_flops_per_eval_all_64_lanes": flops / S, "lane_occupancy": occ, "source": src,
                   "note": "wavefront instructions x 64 lanes; the *_active_lanes figures multiply by the measured share of "
                           "lanes the exec mask had switched on (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU): what the "
                           "hardware usefully did"}
            if occ is not None:
                out.update({"executed_tflops": tf * occ, "executed_frac_of_fp64_valu_peak": tf * occ / FP64_VALU_PEAK_TF,
                            "executed_flops_per_eval": flops / S * occ})
            else:
                out.update({"executed_tflops": None, "executed_frac_of_fp64_valu_peak": None})
            return out
    return {"executed_tflops": None, "executed_frac_of_fp64_valu_peak": None, "source": None}


def qp_algorithmic_bytes(H: int) -> float:
    """What one QP step of a device-resident solve has to move (mpc_amd/csrc/kernels_qp.hip, qp_step_kernel): the packed
    LDL' factor in and out, the gradient, c, the structured dc/dx, the box, the last step and Lagrangian gradient in,
    the multipliers in and out, the new step and Lagrangian gradient out."""
    n, m = 6 * H, 12 * (H + 1)
    tri, nz, mg = n * (n + 1) // 2, 6 * H + 3 * H * (H + 1), m + 2 * n
    return 8.0 * (2 * tri + n + m + nz + 2 * n + 2 * n + 2 * mg + 2 * n)


def solves_roofline(H, n_qp, solved, seconds, solver_seconds=None):
    """The solves leg against its two bounds, from the committed counter passes of its dominant kernel
    (profiles/rNN/qp_roofline.json: FP64 wavefront instructions, HBM bytes and lane occupancy per QP step over real
    recorded QP subproblems, and the kernel's share of the leg's GPU time from the kernel trace of a full-size solve)
    and the LIVE number of QP steps and seconds of this run."""
    for t, src in _profile_entries("qp_roofline.json"):
        if t.get("horizon") != H:
            continue
        share = float(t["share_of_gpu_time"])
        occ = t.get("lane_occupancy") or 1.0
        # inside the solver the leg is one stream of kernels back to back (the rest of `seconds` is host-side tick
        # preparation, unpacking and the gather)
        kernel_s = (solver_seconds or seconds) * share
        flops = float(t["executed_flops_per_qp"]) * n_qp
        traffic = float(t["traffic_bytes_per_qp"]) * n_qp
        tf = flops / kernel_s / 1e12
        return {"dominant_kernel": t["kernel"], "share_of_gpu_time": share, "qp_steps": n_qp,
                "qp_steps_per_solve": n_qp / max(solved, 1.0),
                "kernel_seconds_estimate": kernel_s,
                "executed_flops_per_qp_all_64_lanes": t["executed_flops_per_qp"], "lane_occupancy": t.get("lane_occupancy"),
                "executed_tflops_all_64_lanes": tf, "executed_tflops": tf * occ,
                "frac_of_fp64_valu_peak_all_64_lanes": tf / FP64_VALU_PEAK_TF,
                "frac_of_fp64_valu_peak": tf * occ / FP64_VALU_PEAK_TF, "peak_tflops": FP64_VALU_PEAK_TF,
                "whole_leg_executed_tflops": flops * occ / seconds / 1e12,
                "whole_leg_frac_of_fp64_valu_peak": flops * occ / seconds / 1e12 / FP64_VALU_PEAK_TF,
                "traffic_bytes_per_qp": t["traffic_bytes_per_qp"], "algorithmic_bytes_per_qp": qp_algorithmic_bytes(H),
                "hbm_gbs": traffic / kernel_s / 1e9, "hbm_frac": traffic / kernel_s / 1e9 / HBM_PEAK_GBS,
                "algorithmic_hbm_frac": qp_algorithmic_bytes(H) * n_qp / kernel_s / 1e9 / HBM_PEAK_GBS,
                "bound": "neither peak: barrier-separated recurrences and an HBM-resident NNLS matrix (DESIGN.md 4.4)",
                "source": src}
    return None


ALGORITHMIC_NOTE = ("full-rollout count of SURVEY.md 8(d), (1+6H)(745H+600+30(H+1)) per eval; the prefix kernel shares the "
                    "unperturbed prefix of every perturbed rollout and executes less, so this is a rate of useful work, "
                    "not a fraction of the hardware peak")


def single_tick_latency(device, with_cpu):
    """BASELINE.json configs[0]: ONE controller, default controller_params.yaml (H=1, linear platform motion),
    one `compute_control` call through the drop-in class; the reference-shaped numpy path beside it."""
    from mpc_amd.ur_mpc import GpuURMPC

    mp1 = MpcParams.reference_yaml_defaults()
    mpc = GpuURMPC(mp1, device=device)
    ctx1 = mpc._ctx
    hs1 = make_scenarios(mp1, DeviceKinematics(ctx1), 8, motion="linear", cfg_id=1)
    rows = [i for i in range(8) if hs1.active[i]]

    def dicts(i):
        return ({"position": hs1.q0[i], "velocity": np.zeros(6)},
                {"position": hs1.state0[i, 12:15], "orientation": hs1.state0[i, 15:18]},
                {"position": hs1.target[i, :3], "orientation": hs1.target[i, 3:]},
                {"position": hs1.base[i, 0:3], "orientation": hs1.base[i, 3:6], "linear_velocity": hs1.base[i, 6:9],
                 "angular_velocity": hs1.base[i, 9:12]})

    mpc.compute_control(*dicts(rows[0]))                  # warm (first launch, scipy import paths)
    times = []
    for i in rows:
        one = GpuURMPC(mp1, device=device)
        t0 = time.perf_counter()
        one.compute_control(*dicts(i))
        times.append(time.perf_counter() - t0)
    out = {"ms_per_tick": 1e3 * float(np.median(times)), "ticks": len(times), "horizon": mp1.horizon,
           "what": "GpuURMPC.compute_control, scipy minimize(SLSQP) with device callbacks, one scenario"}
    if with_cpu:
        from oracle.numpy_ref import NumpyRefMPC
        from oracle.oracle import ref_params_from

        cpu = []
        for i in rows:
            ref = NumpyRefMPC(ref_params_from(mp1))
            t0 = time.perf_counter()
            ref.compute_control(*dicts(i))
            cpu.append(time.perf_counter() - t0)
        out["cpu_numpy_ms_per_tick"] = 1e3 * float(np.median(cpu))
    # the same single tick at the benchmark horizon (H=20): scipy's own SLSQP core on the host, callbacks on the device
    mp20 = MpcParams.reference_yaml_defaults(horizon=20)
    m20 = GpuURMPC(mp20, device=device)
    hs20 = make_scenarios(mp20, DeviceKinematics(m20._ctx), 6, motion="linear", cfg_id=1)
    rows20 = [i for i in range(6) if hs20.active[i]][:4]

    def dicts20(i):
        return ({"position": hs20.q0[i], "velocity": np.zeros(6)},
                {"position": hs20.state0[i, 12:15], "orientation": hs20.state0[i, 15:18]},
                {"position": hs20.target[i, :3], "orientation": hs20.target[i, 3:]},
                {"position": hs20.base[i, 0:3], "orientation": hs20.base[i, 3:6], "linear_velocity": hs20.base[i, 6:9],
                 "angular_velocity": hs20.base[i, 9:12]})

    m20.compute_control(*dicts20(rows20[0]))
    t20 = []
    for i in rows20:
        one = GpuURMPC(mp20, device=device)
        t0 = time.perf_counter()
        one.compute_control(*dicts20(i))
        t20.append(time.perf_counter() - t0)
    out["H20_ms_per_tick"] = 1e3 * float(np.median(t20))
    # the same ticks with minimize replaced by the device-resident solver (GpuURMPC(solver="resident"): same controls)
    for key, mp_k, dk, rows_k in (("resident_solver_ms_per_tick", mp1, dicts, rows), ("H20_resident_solver_ms_per_tick", mp20, dicts20, rows20)):
        GpuURMPC(mp_k, device=device, solver="resident").compute_control(*dk(rows_k[0]))
        tr = []
        for i in rows_k:
            one = GpuURMPC(mp_k, device=device, solver="resident")
            t0 = time.perf_counter()
            one.compute_control(*dk(i))
            tr.append(time.perf_counter() - t0)
        out[key] = 1e3 * float(np.median(tr))
    out["H20_note"] = ("same drop-in class at H=20 (scipy's compiled SLSQP core on one host core, every callback served by "
                       "the device); the numpy path's H=20 tick is cpu_baseline.solves_value (about 4 s)")
    return out


def solve_scenarios(ctx, mp, count, rank, sweep_motion=None):
    """The solves leg's scenarios: configs[2]'s angular-motion list, or this rank's rows of the configs[4] sweep."""
    if sweep_motion is not None:
        return make_sweep(mp, DeviceKinematics(ctx), rank * count, count, motion=sweep_motion)
    return make_scenarios(mp, DeviceKinematics(ctx), count, motion="angular", cfg_id=3, goal_jitter=0.0, first=rank * count)


def solve_rate(ctx, mp, sample, threads, rank, world, dist, driver="resident", sweep_motion=None):
    """MPC solves/s on a bounded sample (BASELINE.json's second metric): full SLSQP solves of
    angular-motion scenarios (configs[2] shape), one tick each.  Drivers (identical iterates, bit for bit):
      resident   everything on the GPU (mpc_solve_*): evaluations, BFGS update + QP, and Kraft's O(n + m) body as
                 a kernel over device-side request lists; one host thread enqueues rounds (ResidentSLSQP);
      device_qp  evaluations and BFGS update + QP on the GPU (mpc_qp_step), the body in libmpc_slsqp.so on `threads`
                 host threads per GPU (DeviceQpSLSQP);
      native     the whole SLSQP core on the host threads (NativeBatchedSLSQP), evaluations on the GPU: host-bound.
    Every rank calls this; a rank that fails reports it and all ranks skip together (no hang)."""
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.native_solver import DeviceQpSLSQP, NativeBatchedSLSQP
    from mpc_amd.solver import BatchedMPC

    device_qp = driver != "native"
    if driver == "resident":
        threads = 1

    def all_ok(ok):
        if dist is None:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=ctx.device)
        all_reduce(dist, t, dist.ReduceOp.MIN)       # doubles as the barrier before the timed part
        return bool(t.item() > 0.5)

    count = min(sample, ctx.max_scenarios)
    native, err, hs, bm, dt, table = None, None, None, None, 0.0, None
    try:
        hs = solve_scenarios(ctx, mp, count, rank, sweep_motion)
        if driver == "resident":
            native = ResidentSLSQP(ctx, capacity=count, lanes=1)
        else:
            native = (DeviceQpSLSQP if driver == "device_qp" else NativeBatchedSLSQP)(ctx, threads=threads, capacity=count)
        log("solver up")
        # untimed warm-up: the resident driver takes a whole tick of the same size (the timed tick below then finds torch's
        # caching allocator holding its 0.4 GB of buffers, as every tick after a farm's first does; a fresh BatchedMPC
        # makes it a cold start again: no warm starts, the same solves), the host-bound drivers a small one
        warm = count if driver == "resident" else min(count, 32)
        prep = "resident" if driver == "resident" else "host"    # resident: start points and solutions never leave the device
        BatchedMPC(ctx, solver=native, prep=prep).compute_control(hs.q0[:warm], hs.state0[:warm, 12:18], hs.target[:warm],
                                                                  hs.base[:warm])
        torch.cuda.synchronize()
    except Exception as exc:  # noqa: BLE001 - the evals/s line must survive a host-side problem
        err = repr(exc)
    try:
        log(f"native solver warm, err={err}")
        if not all_ok(err is None):
            return {"error": err or "another rank failed to start its solver"}
        try:
            bm = BatchedMPC(ctx, solver=native, prep=prep)
            t0 = time.perf_counter()
            u = bm.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
            table = gather_result_rows(dist, solve_result_rows(u, bm.last, bm.last_rows), world, rank, ctx.device)
            dt = time.perf_counter() - t0
        except Exception as exc:  # noqa: BLE001
            err = repr(exc)
        if not all_ok(err is None):
            return {"error": err or "another rank failed during its solves"}
    finally:
        if native is not None:
            native.close()
    r = bm.last
    stats = torch.tensor([dt, float(hs.active.sum()), float(r.success.sum()), float(r.n_eval_fd), float(r.n_eval),
                          float(r.n_qp) if device_qp else 0.0], dtype=torch.float64, device=ctx.device)
    if dist is not None:
        mx = stats.clone()
        all_reduce(dist, mx, dist.ReduceOp.MAX)
        all_reduce(dist, stats, dist.ReduceOp.SUM)
        stats[0] = mx[0]
    dt_all, solved, conv, nfd, nf, nqp = (float(v) for v in stats.tolist())
    what = (f"rows of the 32x32 amplitude x phase lattice x 1,024 goal poses sweep, {sweep_motion} platform motion"
            if sweep_motion is not None else "angular-motion scenarios")
    where = {"resident": "whole solves resident on the GPU (mpc_solve_*: Kraft's body, BFGS updates and QP subproblems "
                         "as kernels over device-side request lists), one host thread per GPU enqueuing rounds",
             "device_qp": f"BFGS updates and QP subproblems on the GPU (mpc_qp_step), SLSQP's scalar logic (libmpc_slsqp.so) "
                          f"on {threads} host threads per GPU",
             "native": f"SLSQP cores (libmpc_slsqp.so) on {threads} host threads per GPU"}[driver]
    out = {"value": solved / dt_all, "unit": "solves/s",
           "driver": {"resident": "ResidentSLSQP", "device_qp": "DeviceQpSLSQP", "native": "NativeBatchedSLSQP"}[driver],
           "host_bound": not device_qp, "host_threads_per_gpu": threads,
           "sample": f"{int(solved)} {what} ({count} per GPU), H={mp.horizon}, one tick each, "
                     f"{int(conv)} converged, {int(nfd)} eval_fd + {int(nf)} eval scenario-evals on the GPU, {where}",
           "seconds": dt_all, "mean_iterations": float(r.nit.mean()), "rounds_rank0": r.rounds,
           "rank0_core_seconds_busiest_thread": r.t_core, "rank0_seconds_serving_from_gpu": r.t_serve}
    out["gathered"] = {"rows": int(table.shape[0]), "columns": "f*, u0[6], nit, status",
                       "converged_rows": int((table[:, 8] == 0).sum()), "early_exit_rows": int((table[:, 8] == -1).sum()),
                       "sum_abs_u0": float(np.abs(table[:, 1:7]).sum()),
                       "how": "one all_gather of every rank's [count, 9] block, inside the timed region"}
    if driver == "resident":
        # per GPU: every rank runs the same kernels on its own shard, so the fractions are those of one GPU
        out["roofline"] = solves_roofline(mp.horizon, nqp / world, solved / world, dt_all, r.t_serve)
    if device_qp:
        out["rank0_qp_subproblems_on_gpu"] = int(r.n_qp)
        out["rank0_solved_again_on_host"] = int(r.n_host_fallback)
    if driver == "resident" and getattr(r, "n_free_steps", None) is not None and r.n_qp:
        # QP subproblems whose unconstrained step satisfied every constraint row with a margin (lsq's free step: no
        # transformed matrix, no NNLS; same bits as the long form)
        out["rank0_qp_free_step_share"] = float(r.n_free_steps) / float(r.n_qp)
    return out


def pool_solve_rate(ctx, mp, per_worker, workers):
    """The same sample shape through the process pool over scipy's own compiled core (mpc_amd/pool.py)."""
    from mpc_amd.pool import SolverPool
    from mpc_amd.solver import BatchedMPC

    count = per_worker * workers
    hs = make_scenarios(mp, DeviceKinematics(ctx), count, motion="angular", cfg_id=3, goal_jitter=0.0)
    with SolverPool(ctx, workers, capacity=count) as pool:
        BatchedMPC(ctx, solver=pool).compute_control(hs.q0[:workers], hs.state0[:workers, 12:18], hs.target[:workers], hs.base[:workers])
        bm = BatchedMPC(ctx, solver=pool)
        t0 = time.perf_counter()
        bm.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
        dt = time.perf_counter() - t0
    return {"value": float(hs.active.sum()) / dt, "unit": "solves/s", "workers": workers,
            "sample": f"{int(hs.active.sum())} scenarios, scipy's compiled SLSQP core in {workers} worker processes"}


def rehearse_launch(args, world):
    """No GPU: the ranks rendezvous over gloo, each contributes synthetic [count, 9] rows that depend only on
    the GLOBAL scenario index, the table is gathered as in the solves leg and rank 0 prints a line of the usual
    shape.  Exercises launch_ranks, the rank/shard bookkeeping and gather_result_rows; measures nothing."""
    global _HOST_COLLECTIVES
    _HOST_COLLECTIVES = True
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")
    if os.environ.get("MPC_BENCH_REHEARSE_FAIL_RANK") == str(rank):     # test hook: one rank dies before the gather
        sys.exit(f"rank {rank}: failing on request")
    S = min(args.scenarios or 1000, 4096)

    def rows_of(gidx):
        if args.workload == "sweep":        # [A, phi, goal offset (6), goal index] of the global sweep rows: no kinematics
            from mpc_amd.scenarios import sweep_goal_table, sweep_index, sweep_lattice

            ia, ip, goal = sweep_index(gidx)
            A, phi = sweep_lattice(args.motion)
            return np.concatenate([A[ia][:, None], phi[ip][:, None], sweep_goal_table(args.horizon)[goal, :6],
                                   goal[:, None].astype(float)], axis=1)
        return np.stack([np.sin(gidx * (k + 1.0)) for k in range(9)], axis=1)

    # the sweep rehearsal strides the shards over the whole lattice (rank r starts at row r * 131,072 like the real run)
    per_rank = SWEEP_ROWS // 8 if args.workload == "sweep" else S
    local = rows_of(rank * per_rank + np.arange(S))
    t0 = time.perf_counter()
    table = gather_result_rows(dist, local, world, rank, None)
    dt = time.perf_counter() - t0
    want = np.concatenate([rows_of(r * per_rank + np.arange(S)) for r in range(world)])
    assert np.array_equal(table, want), "gathered table is not the single-process table"
    if rank == 0:
        print(json.dumps({"metric": "launch rehearsal (no GPU, numbers mean nothing)", "value": 0.0, "unit": "evals/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                          "data": "synthetic", "config": {"workload": f"rehearsal ({args.workload})", "scenarios_per_gpu": S,
                                                          "parallelism": f"scenario-dp{world}"},
                          "solves": {"gathered": {"rows": int(table.shape[0]), "checksum": float(table.sum())},
                                     "host_threads_per_gpu": int(os.environ.get("OMP_NUM_THREADS", "0"))}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenarios", type=int, default=0, help="scenarios per GPU (default 65,536; 131,072 for the sweep)")
    ap.add_argument("--workload", default="scenarios", choices=["scenarios", "sweep"],
                    help="scenarios: i.i.d. draws of the scenario family (BASELINE configs[1]-[3]); sweep: configs[4], "
                         "the 32x32 amplitude x phase lattice x 1,024 goal poses, rank r takes rows [r S, (r+1) S)")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--motion", default="linear", choices=["linear", "angular"])
    ap.add_argument("--algo", type=int, default=_ffi.ALGO_AUTO)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--solve-sample", type=int, default=-1,
                    help="scenarios per GPU in the bounded solves/s sample (0: skip; default: --scenarios)")
    ap.add_argument("--solve-workers", type=int, default=0, help="SLSQP host threads per GPU (0: host cores - 1)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: every rank uses device 0 and the process group is gloo, to walk the "
                         "multi-rank code path on a one-GPU box (the numbers mean nothing)")
    ap.add_argument("--only-main", action="store_true",
                    help="run only the timed loop (profiling: every dispatch of the kernel is the benchmark's)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="development / CPU test only: walk the launch + gather path with NO GPU (gloo, synthetic "
                         "per-scenario rows); prints a line whose numbers mean nothing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the ranks ourselves, before anything here touches the GPU
        # (no device count here: that would bring the HIP runtime up in this process; a rank without a GPU fails cleanly)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    if args.rehearse_launch:
        return rehearse_launch(args, world)
    sweep = args.workload == "sweep"
    if not args.scenarios:
        args.scenarios = SWEEP_ROWS // 8 if sweep else 65536
    if args.solve_sample < 0:
        args.solve_sample = args.scenarios
    if sweep and world * args.scenarios > SWEEP_ROWS:
        sys.exit(f"bench.py: the sweep has {SWEEP_ROWS} rows, {world} ranks x {args.scenarios} is more")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: mpc_amd has no CPU path")
    if args.rehearse_on_one_gpu:
        global _HOST_COLLECTIVES
        _HOST_COLLECTIVES = True
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    log("process group up")
    H, S = args.horizon, args.scenarios
    mp = MpcParams.reference_yaml_defaults(horizon=H, algo=args.algo)
    ctx = MpcContext(mp, max_scenarios=S, device=local_rank)
    # this rank's contiguous shard of the global scenario list (rows depend only on the global index)
    if sweep:
        hs = make_sweep(mp, DeviceKinematics(ctx), rank * S, S, motion=args.motion)
    else:
        hs = make_scenarios(mp, DeviceKinematics(ctx), S, motion=args.motion, cfg_id=2, first=rank * S)
    sc, x = to_device(ctx, hs)
    out = EvalResult(*(torch.empty(s, dtype=torch.float64, device=ctx.device)
                       for s in ((S,), (S, mp.n), (S, mp.m), (S, mp.n, mp.m))))

    def step():
        ctx.eval_fd(sc, x, out=out)

    log("scenarios resident")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("warm")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                       # the kernels are launched on torch's current stream
    for _ in range(args.steps):
        step()
    ev1.record()
    gathered = None
    if dist is not None:               # the one exchange of the job: per-scenario results to every rank
        gathered = torch.empty(world * S, dtype=torch.float64, device=ctx.device)
        all_gather(dist, gathered, out.cost)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log("timed loop done")
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        all_reduce(dist, t, dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the timed outputs are finite and the gather is the concatenation of the shards
    assert bool(torch.isfinite(out.cost).all()) and bool(torch.isfinite(out.grad).all())
    if gathered is not None:
        assert torch.equal(gathered[rank * S:(rank + 1) * S], out.cost)

    solves, workers = None, 1
    if not args.only_main and args.solve_sample > 0:        # every rank takes part: one pool per GPU
        from mpc_amd.pool import host_core_budget

        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        workers = args.solve_workers or max(1, min(15, host_core_budget(local_world) - 1))
        log(f"solves leg (host threads for the comparison drivers: {workers})")
        solves = solve_rate(ctx, mp, args.solve_sample, workers, rank, world, dist,
                            sweep_motion=args.motion if sweep else None)
        log("solves leg done")

    if rank == 0:
        total_evals = world * S * args.steps
        value = total_evals / elapsed
        abytes, aflops = algorithmic_bytes(H), algorithmic_flops(H)
        achieved = abytes * S / (kernel_ms * 1e-3) / 1e9
        algo_name = "prefix" if ctx.params.algo != _ffi.ALGO_ROLLOUT else "rollout"
        traffic, traffic_src = measured_traffic(H, S, algo_name)
        if traffic is None and H > 40:       # long horizons: the rollout kernel and the expansion are two kernels
            parts = [measured_traffic(H, S, algo_name, v) for v in ("rollout", "expand")]
            if all(p[0] is not None for p in parts):
                traffic, traffic_src = sum(p[0] for p in parts), parts[0][1]
        line = {
            "metric": "objective+gradient evals/sec (f, c, FD grad f, dense FD dc/dx per scenario)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"sweep of platform amplitude x phase (32x32 lattice) x 1,024 goal poses (BASELINE configs[4]), "
                                    f"{args.motion} platform motion, rows [r*{S}, (r+1)*{S}) on rank r, H={H}, objective+FD "
                                    f"gradient+dense FD constraint Jacobian, resident in HBM" if sweep else
                                    f"{args.motion}-platform-motion scenarios, H={H}, objective+FD gradient+dense FD "
                                    f"constraint Jacobian, {S} scenarios per GPU resident in HBM (north_star target size)"),
                       "horizon": H, "scenarios_per_gpu": S, "n_vars": mp.n, "n_constraints": mp.m,
                       "algo": algo_name,
                       "parallelism": f"scenario-dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": abytes * S,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_eval": abytes,
                         "fp64_valu": dict(executed_fp64(H, S, "jac", kernel_ms), peak_tflops=FP64_VALU_PEAK_TF,
                                           algorithmic_tflops=aflops * S / (kernel_ms * 1e-3) / 1e12,
                                           algorithmic_flops_per_eval=aflops, algorithmic_note=ALGORITHMIC_NOTE)},
        }
        # compute-side view: same evals without the dense dc/dx stream (gradient + constraints only)
        extras = not args.only_main
        out2 = EvalResult(out.cost, out.grad, out.con, None)
        for _ in range(2 if extras else 0):
            ctx.eval_fd(sc, x, want_jac=False, out=out2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps if extras else 0):
            ctx.eval_fd(sc, x, want_jac=False, out=out2)
        e1.record()
        torch.cuda.synchronize()
        ms_nj = e0.elapsed_time(e1) / args.steps
        if extras:
            line["no_dense_jacobian"] = dict({"evals_per_s": S / (ms_nj * 1e-3), "kernel_ms": ms_nj,
                                              "algorithmic_tflops": aflops * S / (ms_nj * 1e-3) / 1e12},
                                             **executed_fp64(H, S, "nojac", ms_nj))
        if extras and ctx.nz_supported():
            # same derivative information with dc/dx in its structured form (include/mpc_amd.h, mpc_eval_fd_nz):
            # 11 KB instead of 242 KB per eval at H=20, so this figure is bound by the FP64 pipes, not HBM
            z = ctx.eval_fd_nz(sc, x)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.steps):
                ctx.eval_fd_nz(sc, x, out=z)
            e1.record()
            torch.cuda.synchronize()
            ms_nz = e0.elapsed_time(e1) / args.steps
            line["structured_jacobian"] = dict({"evals_per_s": S / (ms_nz * 1e-3), "kernel_ms": ms_nz,
                                                "bytes_out_per_eval": 8 * (1 + mp.n + mp.m + ctx.nz_count),
                                                "algorithmic_tflops": aflops * S / (ms_nz * 1e-3) / 1e12},
                                               **executed_fp64(H, S, "nz", ms_nz))
            del z
        if extras and S > 4096:   # configs[1] of BASELINE.json: 4,096 scenarios on one GPU
            sc4, x4 = sc.slice(0, 4096), x[:4096]
            o4 = EvalResult(out.cost[:4096], out.grad[:4096], out.con[:4096], out.con_jac[:4096])
            for _ in range(3):
                ctx.eval_fd(sc4, x4, out=o4)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.steps):
                ctx.eval_fd(sc4, x4, out=o4)
            e1.record()
            torch.cuda.synchronize()
            line["evals_per_s_4096_scenarios"] = 4096 / (e0.elapsed_time(e1) / args.steps * 1e-3)
        if solves is not None:
            line["solves"] = solves
            if extras and world == 1 and not sweep and "error" not in solves:
                try:        # 4,096 scenarios (configs[1]'s batch size) through the same driver: latency-bound rounds
                    line["solves_4096_scenarios"] = solve_rate(ctx, mp, 4096, 1, 0, 1, None)
                except Exception as exc:  # noqa: BLE001 - a comparison figure, not the metric
                    line["solves_4096_scenarios"] = {"error": repr(exc)}
                try:        # SLSQP's scalar logic on host threads, QP on the GPU: last round's default driver
                    line["solves_device_qp_host_logic"] = solve_rate(ctx, mp, args.solve_sample, workers, 0, 1, None,
                                                                     driver="device_qp")
                except Exception as exc:  # noqa: BLE001 - a comparison figure, not the metric
                    line["solves_device_qp_host_logic"] = {"error": repr(exc)}
                try:        # the host-only driver on a smaller sample: what the GPU-side QP buys
                    line["solves_host_core"] = solve_rate(ctx, mp, 256 * workers, workers, 0, 1, None, driver="native")
                except Exception as exc:  # noqa: BLE001 - a comparison figure, not the metric
                    line["solves_host_core"] = {"error": repr(exc)}
                try:
                    line["solves_scipy_core_pool"] = pool_solve_rate(ctx, mp, 32, workers)
                except Exception as exc:  # noqa: BLE001 - a comparison figure, not the metric
                    line["solves_scipy_core_pool"] = {"error": repr(exc)}
        # rank 0 alone, after the timed legs, also when N > 1: the other ranks wait at the closing barrier
        if extras and not sweep:
            line["single_tick_default_yaml"] = single_tick_latency(local_rank, not args.no_cpu_baseline)
        if extras and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(mp, hs)
            line["speedup_vs_cpu_core"] = value / world / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
