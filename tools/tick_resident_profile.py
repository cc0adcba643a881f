"""Dev tool: host profile of one GpuURMPC(solver="resident") tick (default YAML, H = 1, and H = 20).
usage: tick_resident_profile.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats, time
import numpy as np
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios
from mpc_amd.ur_mpc import GpuURMPC

for H in (1, 20):
    mp = MpcParams.reference_yaml_defaults(horizon=H)
    mpc = GpuURMPC(mp, solver="resident")
    hs = make_scenarios(mp, DeviceKinematics(mpc._ctx), 12, motion="linear", cfg_id=1)
    rows = [i for i in range(12) if hs.active[i]]
    def dicts(i):
        return ({"position": hs.q0[i], "velocity": np.zeros(6)}, {"position": hs.state0[i, 12:15], "orientation": hs.state0[i, 15:18]},
                {"position": hs.target[i, :3], "orientation": hs.target[i, 3:]},
                {"position": hs.base[i, 0:3], "orientation": hs.base[i, 3:6], "linear_velocity": hs.base[i, 6:9], "angular_velocity": hs.base[i, 9:12]})
    mpc.compute_control(*dicts(rows[0]))
    ts = []
    for i in rows[:6]:
        mpc.previous_solution = None
        t0 = time.perf_counter(); mpc.compute_control(*dicts(i)); ts.append(time.perf_counter() - t0)
    print(f"H={H}: median {1e3*np.median(ts):.3f} ms, min {1e3*min(ts):.3f} ms", flush=True)
    pr = cProfile.Profile(); mpc.previous_solution = None
    pr.enable(); mpc.compute_control(*dicts(rows[1])); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
