"""Dev tool: where one batched tick (BatchedMPC.compute_control, 65,536 controllers) spends its time outside the solver."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats, time
import numpy as np, torch
from mpc_amd.batch import MpcContext
from mpc_amd.device_solver import ResidentSLSQP
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios
from mpc_amd.solver import BatchedMPC

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prep = sys.argv[2] if len(sys.argv) > 2 else "host"
mp = MpcParams.reference_yaml_defaults(horizon=20)
ctx = MpcContext(mp, N, 0)
hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3, goal_jitter=0.0)
with ResidentSLSQP(ctx, capacity=N) as solver:
    BatchedMPC(ctx, solver=solver, prep=prep).compute_control(hs.q0[:64], hs.state0[:64, 12:18], hs.target[:64], hs.base[:64])
    bm = BatchedMPC(ctx, solver=solver, prep=prep)
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable()
    bm.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
    pr.disable(); dt = time.perf_counter() - t0
    print(f"N={N} prep={prep}: tick {dt:.3f}s, solver {bm.last.t_wall:.3f}s, outside {dt - bm.last.t_wall:.3f}s")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
