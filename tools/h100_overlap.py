"""Dev tool: which part of the H=100 expansion slows the rollouts when both share the chip?
Runs rollouts (no dc/dx, or structured dc/dx) on one stream beside the expansion, or a plain zero fill, on another."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd.batch import MpcContext
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device
H, N = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mp = MpcParams.reference_yaml_defaults(horizon=H, algo=2)
ctx = MpcContext(mp, N, 0)
sc, x = to_device(ctx, make_scenarios(mp, DeviceKinematics(ctx), N, "linear", 2))
z = ctx.eval_fd_nz(sc, x)
z2 = ctx.eval_fd_nz(sc, x)
g = ctx.eval_fd(sc, x, want_jac=False)
jac = ctx.expand_jacobian(z.con_jac_nz)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def R_nojac(): ctx.eval_fd(sc, x, want_jac=False, out=g)
def R_nz(): ctx.eval_fd_nz(sc, x, out=z2)
def E(): ctx.expand_jacobian(z.con_jac_nz, out=jac)
def Z(): jac.zero_()

def timed(fa, fb, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); ea = torch.cuda.Event(enable_timing=True); eb = torch.cuda.Event(enable_timing=True)
        e0.record()
        sa.wait_event(e0); sb.wait_event(e0)
        if fa:
            with torch.cuda.stream(sa): fa(); ea.record()
        if fb:
            with torch.cuda.stream(sb): fb(); eb.record()
        torch.cuda.synchronize()
        ta = e0.elapsed_time(ea) if fa else 0.0
        tb = e0.elapsed_time(eb) if fb else 0.0
        best = min(best, max(ta, tb))
        last = (ta, tb)
    return best, last

for name, fa, fb in [("R_nojac alone", R_nojac, None), ("R_nz alone", R_nz, None), ("E alone", None, E), ("Z alone", None, Z),
                     ("R_nojac || E", R_nojac, E), ("R_nz || E", R_nz, E), ("R_nojac || Z", R_nojac, Z), ("R_nz || Z", R_nz, Z)]:
    best, last = timed(fa, fb)
    print(f"{name:16s}: {best:7.3f} ms   (stream a {last[0]:7.3f}, stream b {last[1]:7.3f})", flush=True)
