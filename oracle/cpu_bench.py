"""CPU-baseline helper for bench.py's `cpu_baseline` leg (test/measurement infrastructure, never the
product): times the reference-shaped numpy path (oracle/numpy_ref.py) in worker processes."""
from __future__ import annotations

import time


def timed_evals(args):
    """One worker: full evals (f, c, scipy-style FD grad f and dc/dx) of its scenarios until the budget
    is spent.  args = (RefParams, rows dict of numpy arrays, budget seconds) -> (evals done, seconds)."""
    import numpy as np

    from oracle.numpy_ref import NumpyRefMPC, base_traj_from_array

    rp, rows, budget = args
    n = rows["x"].shape[0]
    t0 = time.perf_counter()
    done = 0
    while (time.perf_counter() - t0) < budget:
        i = done % n
        ref = NumpyRefMPC(rp)
        ref.J_cached = rows["Jc"][i].reshape(6, 6)
        tgt = {"position": rows["target"][i, :3], "orientation": rows["target"][i, 3:]}
        ref.fd_eval(rows["x"][i], rows["state0"][i], tgt, base_traj_from_array(rows["base_traj"][i]), fast_constraint=False)
        done += 1
    return done, time.perf_counter() - t0


def all_cores_evals_per_s(rp, rows, cores: int, budget: float = 8.0):
    """The same path on `cores` processes at once (scenarios split evenly): evals/s of the box."""
    import multiprocessing as mp

    n = rows["x"].shape[0]
    per = max(1, n // cores)
    parts = [{k: v[(w * per) % n:(w * per) % n + per] for k, v in rows.items()} for w in range(cores)]
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(timed_evals, [(rp, p, 0.2) for p in parts])           # warm the workers (imports)
        t0 = time.perf_counter()
        out = pool.map(timed_evals, [(rp, p, budget) for p in parts])
        wall = time.perf_counter() - t0
    return sum(d for d, _ in out) / wall, wall
