"""Time the batched solve: one process, then pools of W worker processes (96 scenarios per worker).
usage: solve_time.py N W1,W2,...   (MPC_AMD_SOLVE_PROFILE=1 adds a per-phase split of the device calls)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np

PER_WORKER = 96


def main():
    from mpc_amd.batch import MpcContext
    from mpc_amd.params import MpcParams
    from mpc_amd.pool import SolverPool
    from mpc_amd.scenarios import DeviceKinematics, make_scenarios
    from mpc_amd.solver import BatchedMPC

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    Ws = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8]
    M = PER_WORKER * max(Ws)
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    ctx = MpcContext(mp, max(N, M), 0)
    hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3, goal_jitter=0.0)
    bm = BatchedMPC(ctx)
    t0 = time.perf_counter(); bm.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base); dt = time.perf_counter() - t0
    r = bm.last
    print(f"N={N} {dt:.2f}s  {hs.active.sum()/dt:.1f} solves/s core={r.t_core:.2f}s serve={r.t_serve:.2f}s rounds={r.rounds} "
          f"nfd={r.n_eval_fd} nf={r.n_eval} conv={r.success.sum()} prof={ {k: round(v, 2) for k, v in bm.solver.profile.items()} }", flush=True)

    from mpc_amd.scenarios import to_device
    hs = make_scenarios(mp, DeviceKinematics(ctx), M, motion="angular", cfg_id=3, goal_jitter=0.0)
    x0 = np.tile(hs.u_init, (1, 20))
    sc_all, _ = to_device(ctx, hs)
    for W in Ws:
        t0 = time.perf_counter()
        with SolverPool(ctx, W, PER_WORKER * W) as pool:
            t1 = time.perf_counter()
            pool.solve(sc_all.slice(0, W), x0[:W])   # warm
            t2 = time.perf_counter()
            k = PER_WORKER * W
            r = pool.solve(sc_all.slice(0, k), x0[:k])
            t3 = time.perf_counter()
            reg = pool._registered
        print(f"pool W={W} M={k}: start {t1-t0:.1f}s warm {t2-t1:.2f}s solve {t3-t2:.2f}s -> {k/(t3-t2):.1f} solves/s; "
              f"core_max={r.t_core:.2f} server_busy={r.t_serve:.2f} conv={int(r.success.sum())} pinned_shm={reg}", flush=True)
    from mpc_amd.native_solver import NativeBatchedSLSQP
    for W in Ws:
        k = PER_WORKER * W
        with NativeBatchedSLSQP(ctx, threads=W, capacity=k) as native:
            native.solve(sc_all.slice(0, W), x0[:W])   # warm
            t2 = time.perf_counter()
            r = native.solve(sc_all.slice(0, k), x0[:k])
            t3 = time.perf_counter()
        print(f"native T={W} M={k}: solve {t3-t2:.2f}s -> {k/(t3-t2):.1f} solves/s; core_max={r.t_core:.2f} serve={r.t_serve:.2f} "
              f"rounds={r.rounds} conv={int(r.success.sum())}", flush=True)
    from mpc_amd.native_solver import DeviceQpSLSQP
    for M2 in (PER_WORKER * max(Ws), 4 * PER_WORKER * max(Ws)):
        if M2 > M:
            hs2 = make_scenarios(mp, DeviceKinematics(ctx), M2, motion="angular", cfg_id=3, goal_jitter=0.0)
            ctx2 = MpcContext(mp, M2, 0)
            sc2, _ = to_device(ctx2, hs2)
            x02 = np.tile(hs2.u_init, (1, 20))
        else:
            ctx2, sc2, x02 = ctx, sc_all.slice(0, M2), x0[:M2]
        with DeviceQpSLSQP(ctx2, threads=max(Ws), capacity=M2) as dev:
            dev.solve(sc2.slice(0, 8), x02[:8])   # warm
            t2 = time.perf_counter()
            r = dev.solve(sc2, x02)
            t3 = time.perf_counter()
        print(f"device-QP T={max(Ws)} M={M2}: solve {t3-t2:.2f}s -> {M2/(t3-t2):.1f} solves/s; core_sum={r.t_core:.2f} evals={r.t_serve:.2f} "
              f"qp={r.t_qp:.2f} ({r.n_qp} QPs) rounds={r.rounds} conv={int(r.success.sum())} host_fallback={r.n_host_fallback}", flush=True)


if __name__ == "__main__":
    main()
