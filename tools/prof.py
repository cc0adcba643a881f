"""Dev tool: a few launches of one evaluation kernel for rocprofv3 counter passes (every dispatch identical).
usage: prof.py H N algo variant     variant: jac (dense dc/dx) | nojac (gradient only) | nz (structured dc/dx)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd.batch import MpcContext
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
algo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
variant = sys.argv[4] if len(sys.argv) > 4 else "jac"
mp = MpcParams.reference_yaml_defaults(horizon=H, algo=algo)
ctx = MpcContext(mp, N, 0)
hs = make_scenarios(mp, DeviceKinematics(ctx), N, "linear", 2)
sc, x = to_device(ctx, hs)
run = (lambda out=None: ctx.eval_fd_nz(sc, x, out=out)) if variant == "nz" else \
      (lambda out=None: ctx.eval_fd(sc, x, want_jac=(variant == "jac"), out=out))
out = run()
for _ in range(3):
    run(out)
torch.cuda.synchronize()
