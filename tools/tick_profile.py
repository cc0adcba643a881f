"""Dev tool: where the host time of one batched compute_control tick goes (cProfile), resident driver.
usage: tick_profile.py [N] [host|device|resident]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats, time
import numpy as np, torch
from mpc_amd.batch import MpcContext
from mpc_amd.device_solver import ResidentSLSQP
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios
from mpc_amd.solver import BatchedMPC

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prep = sys.argv[2] if len(sys.argv) > 2 else "resident"
mp = MpcParams.reference_yaml_defaults(horizon=20)
ctx = MpcContext(mp, N, 0)
hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3, goal_jitter=0.0)
with ResidentSLSQP(ctx, capacity=N) as solver:
    BatchedMPC(ctx, solver=solver, prep=prep).compute_control(hs.q0[:32], hs.state0[:32, 12:18], hs.target[:32], hs.base[:32])
    torch.cuda.synchronize()
    bm = BatchedMPC(ctx, solver=solver, prep=prep)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    u = bm.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"N={N}: tick {dt:.3f} s, solver {bm.last.t_wall:.3f} s, host around it {dt - bm.last.t_wall:.3f} s")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
