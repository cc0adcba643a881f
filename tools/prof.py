"""Dev tool: a few eval_fd launches for rocprofv3 counter passes (every dispatch identical)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import torch
from mpc_amd.batch import MpcContext, EvalResult
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
algo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
jac = (sys.argv[4] != "nojac") if len(sys.argv) > 4 else True
mp = MpcParams.reference_yaml_defaults(horizon=H, algo=algo)
ctx = MpcContext(mp, N, 0)
hs = make_scenarios(mp, DeviceKinematics(ctx), N, "linear", 2)
sc, x = to_device(ctx, hs)
out = ctx.eval_fd(sc, x, want_jac=jac)
for _ in range(3):
    ctx.eval_fd(sc, x, want_jac=jac, out=out)
torch.cuda.synchronize()
