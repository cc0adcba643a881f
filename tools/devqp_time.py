"""Dev tool: whole-solve rate of DeviceQpSLSQP at the given batch sizes.  `python tools/devqp_time.py 5760 23040`"""
import os, sys; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import time
import numpy as np


def main():
    from mpc_amd.batch import MpcContext
    from mpc_amd.native_solver import DeviceQpSLSQP
    from mpc_amd.params import MpcParams
    from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device
    sizes = [int(a) for a in sys.argv[1:]] or [5760]
    threads = int(os.environ.get("THREADS", "15"))
    lanes = int(os.environ.get("LANES", "4"))
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    for M in sizes:
        ctx = MpcContext(mp, M, 0)
        hs = make_scenarios(mp, DeviceKinematics(ctx), M, motion="angular", cfg_id=3, goal_jitter=0.0)
        sc, _ = to_device(ctx, hs)
        x0 = np.tile(hs.u_init, (1, 20))
        with DeviceQpSLSQP(ctx, threads=threads, capacity=M, lanes=lanes) as dev:
            dev.solve(sc.slice(0, 8), x0[:8])   # warm
            t0 = time.perf_counter()
            r = dev.solve(sc, x0)
            dt = time.perf_counter() - t0
        print(f"device-QP T={threads} lanes={lanes} M={M}: solve {dt:.2f}s -> {M/dt:.1f} solves/s; core_sum={r.t_core:.2f} evals={r.t_serve:.2f} "
              f"qp={r.t_qp:.2f} ({r.n_qp} QPs) rounds={r.rounds} conv={int(r.success.sum())} host_fallback={r.n_host_fallback}", flush=True)
        del ctx, sc


if __name__ == "__main__":
    main()
