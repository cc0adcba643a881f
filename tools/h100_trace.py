"""Dev tool: a few H=100 eval_fd launches for a rocprofv3 --kernel-trace timeline (do the two streams overlap?)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd.batch import MpcContext
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device
H, N = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mp = MpcParams.reference_yaml_defaults(horizon=H, algo=2)
ctx = MpcContext(mp, N, 0)
sc, x = to_device(ctx, make_scenarios(mp, DeviceKinematics(ctx), N, "linear", 2))
out = ctx.eval_fd(sc, x, want_jac=True)
torch.cuda.synchronize()
for _ in range(3):
    ctx.eval_fd(sc, x, want_jac=True, out=out)
torch.cuda.synchronize()
