"""BASELINE.json configs[2] at full size: N angular-motion scenarios, H=20, one full SLSQP solve each,
through the solver pool.  usage: solve_full.py [N] [workers] [out.json] [device_qp|native|pool]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import time
import numpy as np


def main():
    import torch
    from mpc_amd.batch import MpcContext
    from mpc_amd.params import MpcParams
    from mpc_amd.pool import SolverPool, host_core_budget
    from mpc_amd.scenarios import DeviceKinematics, make_scenarios
    from mpc_amd.solver import BatchedMPC

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    W = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, min(15, host_core_budget() - 1))
    out = sys.argv[3] if len(sys.argv) > 3 else None
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    ctx = MpcContext(mp, N, 0)
    t0 = time.perf_counter()
    hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3)
    t_gen = time.perf_counter() - t0
    print(f"{N} scenarios generated in {t_gen:.1f} s; {W} workers", flush=True)
    driver = sys.argv[4] if len(sys.argv) > 4 else "native"
    if driver in ("native", "device_qp"):
        from mpc_amd.native_solver import DeviceQpSLSQP, NativeBatchedSLSQP

        with (NativeBatchedSLSQP if driver == "native" else DeviceQpSLSQP)(ctx, threads=W, capacity=N) as native:
            BatchedMPC(ctx, solver=native).compute_control(hs.q0[:64], hs.state0[:64, 12:18], hs.target[:64], hs.base[:64])  # warm
            bm = BatchedMPC(ctx, solver=native)
            t0 = time.perf_counter()
            u = bm.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
            dt = time.perf_counter() - t0
    else:
        with SolverPool(ctx, W, capacity=N) as pool:
            solve = pool.solve
            pool.solve = lambda sc, x0: solve(sc, x0, progress=lambda t, nf, ng, done: print(
                f"  {t:6.1f} s: {nf} eval + {ng} eval_fd scenario-evals served, {done} workers done", flush=True), progress_every=20.0)
            bm = BatchedMPC(ctx, solver=pool)
            t0 = time.perf_counter()
            u = bm.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
            dt = time.perf_counter() - t0
    r = bm.last
    active = int(hs.active.sum())
    res = {"scenarios": N, "active": active, "driver": driver, "workers": W, "seconds": dt, "solves_per_s": active / dt,
           "converged": int(r.success.sum()), "status_counts": {int(k): int(v) for k, v in zip(*np.unique(r.status, return_counts=True))},
           "mean_iterations": float(r.nit.mean()), "max_iterations": int(r.nit.max()), "n_eval": r.n_eval, "n_eval_fd": r.n_eval_fd,
           "max_core_seconds_per_worker": r.t_core, "server_busy_seconds": r.t_serve, "scenario_generation_seconds": t_gen,
           "max_abs_control": float(np.abs(u).max()), "all_finite": bool(np.isfinite(u).all())}
    if driver == "device_qp":
        res["qp_subproblems_on_gpu"] = int(r.n_qp)
        res["solved_again_on_host"] = int(r.n_host_fallback)
    print(json.dumps(res), flush=True)
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
