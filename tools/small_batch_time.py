"""Dev tool: resident solves of small batches (1, 16, 256, 4096 scenarios at H = 20): seconds per solve and solves/s.
usage: small_batch_time.py [sizes ...] [--rounds | --team]   (force the lock-step rounds / mpc_solve_run)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np, torch
from mpc_amd.batch import MpcContext
from mpc_amd.device_solver import ResidentSLSQP
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

args = [a for a in sys.argv[1:] if not a.startswith("--")]
team_max = 0 if "--rounds" in sys.argv else (1 << 30 if "--team" in sys.argv else None)     # default: the driver's own choice
sizes = [int(a) for a in args] or [1, 16, 256, 4096]
mp = MpcParams.reference_yaml_defaults(horizon=20)
cap = max(sizes)
ctx = MpcContext(mp, cap, 0)
hs = make_scenarios(mp, DeviceKinematics(ctx), cap + 64, motion="angular", cfg_id=3, goal_jitter=0.0)
sc_all, _ = to_device(ctx, hs)
with ResidentSLSQP(ctx, capacity=cap, team_max=team_max) as solver:
    for N in sizes:
        sc = sc_all.slice(0, N)
        x0 = hs.x[:N]
        solver.solve(sc, x0)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = solver.solve(sc, x0); ts.append(time.perf_counter() - t0)
        print(f"N={N}: {1e3*min(ts):.3f} ms per solve ({N/min(ts):.0f} solves/s), nit max {int(r.nit.max())}, "
              f"converged {int(r.success.sum())}", flush=True)
