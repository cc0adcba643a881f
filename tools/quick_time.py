"""Quick device timing of eval_fd (dev tool, not the benchmark contract)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time
import numpy as np, torch
from mpc_amd.batch import MpcContext, EvalResult
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
algo = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mp = MpcParams.reference_yaml_defaults(horizon=H, algo=algo)
ctx = MpcContext(mp, N, 0)
t0 = time.time()
hs = make_scenarios(mp, DeviceKinematics(ctx), N, "linear", 2)
print("scenario gen s", time.time() - t0, flush=True)
sc, x = to_device(ctx, hs)
for want_jac in (True, False):
    out = ctx.eval_fd(sc, x, want_jac=want_jac)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 5
    e0.record()
    for _ in range(K):
        ctx.eval_fd(sc, x, want_jac=want_jac, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    bytes_out = N * (1 + mp.n + mp.m + (mp.n * mp.m if want_jac else 0)) * 8
    print(f"H={H} N={N} algo={algo} jac={want_jac}: {ms:.3f} ms/step, {N/ms*1e3:.3e} evals/s, out {bytes_out/ms/1e6:.1f} GB/s", flush=True)

z = ctx.eval_fd_nz(sc, x)
torch.cuda.synchronize()
e0.record()
for _ in range(K):
    ctx.eval_fd_nz(sc, x, out=z)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print(f"H={H} N={N} structured dc/dx: {ms:.3f} ms/step, {N/ms*1e3:.3e} evals/s", flush=True)

if N * mp.n * mp.m * 8 < 120e9:
    jac = ctx.expand_jacobian(z.con_jac_nz)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(K):
        ctx.expand_jacobian(z.con_jac_nz, out=jac)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print(f"H={H} N={N} expand alone: {ms:.3f} ms/step, {N*mp.n*mp.m*8/ms/1e6:.1f} GB/s", flush=True)
