"""Dev tool: ONE controller, H=20 -- the tick through the drop-in class (scipy's core on the host) beside the same tick
through BatchedMPC with the library's own SLSQP core (libmpc_slsqp.so, bit-identical to scipy's, ~5x faster)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np, torch
from mpc_amd.batch import MpcContext
from mpc_amd.native_solver import NativeBatchedSLSQP
from mpc_amd.device_solver import ResidentSLSQP
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios
from mpc_amd.solver import BatchedMPC
from mpc_amd.ur_mpc import GpuURMPC

H = 20
mp = MpcParams.reference_yaml_defaults(horizon=H)
ctx = MpcContext(mp, 8, 0)
hs = make_scenarios(mp, DeviceKinematics(ctx), 8, motion="linear", cfg_id=1)
rows = [i for i in range(8) if hs.active[i]][:4]
def dicts(i):
    return ({"position": hs.q0[i], "velocity": np.zeros(6)},
            {"position": hs.state0[i, 12:15], "orientation": hs.state0[i, 15:18]},
            {"position": hs.target[i, :3], "orientation": hs.target[i, 3:]},
            {"position": hs.base[i, 0:3], "orientation": hs.base[i, 3:6], "linear_velocity": hs.base[i, 6:9],
             "angular_velocity": hs.base[i, 9:12]})
GpuURMPC(mp, device=0).compute_control(*dicts(rows[0]))
ref = {}
t = []
for i in rows:
    one = GpuURMPC(mp, device=0)
    t0 = time.perf_counter(); ref[i] = one.compute_control(*dicts(i)); t.append(time.perf_counter() - t0)
print(f"GpuURMPC (scipy core): {1e3*np.median(t):.2f} ms per tick")
for name, make in (("native core, 1 thread", lambda: NativeBatchedSLSQP(ctx, threads=1, capacity=8, groups=1)),
                   ("resident", lambda: ResidentSLSQP(ctx, capacity=8))):
    solver = make()
    BatchedMPC(ctx, solver=solver).compute_control(hs.q0[rows[:1]], hs.state0[rows[:1], 12:18], hs.target[rows[:1]], hs.base[rows[:1]])
    t = []; err = 0.0
    for i in rows:
        bm = BatchedMPC(ctx, solver=solver)
        r = [i]
        t0 = time.perf_counter()
        u = bm.compute_control(hs.q0[r], hs.state0[r, 12:18], hs.target[r], hs.base[r])
        t.append(time.perf_counter() - t0)
        err = max(err, float(np.abs(np.asarray(u)[0] - np.asarray(ref[i])).max()))
    print(f"BatchedMPC N=1, {name}: {1e3*np.median(t):.2f} ms per tick, max |u - u_scipy| = {err:.2e}")
