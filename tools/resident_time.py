"""Time whole batched solves: ResidentSLSQP (everything on the device) beside DeviceQpSLSQP (host threads keep
SLSQP's scalar logic).  usage: resident_time.py N1,N2,... [lanes] [rounds_per_check] [--no-devqp] [--hash]
--hash: also print a SHA-256 over the solutions, f, iteration counts and exit modes (two builds of the library that print
the same hash solved every problem with the same bits)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import torch


def main():
    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.native_solver import DeviceQpSLSQP
    from mpc_amd.params import MpcParams
    from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

    Ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096]
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rpc = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    for N in Ns:
        ctx = MpcContext(mp, N, 0)
        hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3, goal_jitter=0.0)
        keep = np.nonzero(hs.active)[0]
        sc_all, _ = to_device(ctx, hs)
        from mpc_amd.batch import ScenarioBatch
        sc = ctx.gather(sc_all, torch.from_numpy(keep.astype(np.int32)).to(ctx.device), ScenarioBatch.empty(ctx.device, len(keep), 20, False))
        x0 = np.tile(hs.u_init, (1, 20))[keep]
        M = len(keep)
        with ResidentSLSQP(ctx, capacity=M, lanes=lanes, rounds_per_check=rpc) as dev:
            dev.solve(sc.slice(0, 8), x0[:8])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = dev.solve(sc, x0)
            dt = time.perf_counter() - t0
        print(f"resident N={M} lanes={len(dev.lanes) if dev.lanes else lanes} rpc={rpc}: {dt:.3f}s -> {M/dt:.0f} solves/s; rounds={r.rounds} "
              f"qps={r.n_qp} nf={r.n_eval} nfd={r.n_eval_fd} conv={int(r.success.sum())} fallback={r.n_host_fallback}", flush=True)
        if "--hash" in sys.argv:
            import hashlib
            h = hashlib.sha256()
            for a in (r.x, r.fun, r.nit.astype(np.int64), r.status.astype(np.int64)):
                h.update(np.ascontiguousarray(a).tobytes())
            print(f"hash N={M}: {h.hexdigest()} nit sum {int(r.nit.sum())}", flush=True)
        if "--no-devqp" not in sys.argv:
            with DeviceQpSLSQP(ctx, threads=15, capacity=M) as dq:
                dq.solve(sc.slice(0, 8), x0[:8])
                t0 = time.perf_counter()
                r2 = dq.solve(sc, x0)
                dt2 = time.perf_counter() - t0
            same = np.array_equal(r.x, r2.x) and np.array_equal(r.nit, r2.nit)
            print(f"device-QP N={M} T=15: {dt2:.3f}s -> {M/dt2:.0f} solves/s; rounds={r2.rounds} identical={same}", flush=True)


if __name__ == "__main__":
    main()
