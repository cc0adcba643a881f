"""Dev tool: mpc_qp_step with and without the BFGS update on QPs recorded from host solves (H=20), tiled to a
large batch; the update's inputs (last step, multipliers, old Lagrangian gradient) are synthetic."""
import os, sys; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import ctypes, time
import numpy as np, torch
from helpers import host_scenarios, params_for
from mpc_amd import native_solver
from mpc_amd.batch import MpcContext
from mpc_amd.native_solver import NativeBatch
from mpc_amd.slsqp_state import nz_scatter_index

H, N = 20, 6
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = native_solver.load()
mp = params_for(H)
co, hs = host_scenarios(mp, N, motion="angular", cfg_id=3, goal_jitter=0.0)
n, m = mp.n, mp.m
lb = np.array([b[0] for b in mp.bounds()]); ub = np.array([b[1] for b in mp.bounds()])
scatter = nz_scatter_index(H); ldl = (n + 1) * n // 2 + 1
dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
FN = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip)
builtin = FN(("mpc_slsqp_lsq_builtin", lib)); rec = []; t_host = [0.0]
def hook(m_, meq, n_, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode):
    arr = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
    inp = (arr(l, ldl), arr(g, n), arr(b, m_), arr(a, (n + 1) * la)[scatter], arr(xl, n), arr(xu, n))
    t0 = time.perf_counter(); builtin(m_, meq, n_, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode); t_host[0] += time.perf_counter() - t0
    rec.append(inp + (arr(x, n),))
cb = FN(hook); lib.mpc_slsqp_set_lsq(ctypes.cast(cb, ctypes.c_void_p))
def rows(idx, X):
    d = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}; d["x"] = np.ascontiguousarray(X); return d
b = NativeBatch(H, N, threads=1); b.begin(np.tile(hs.u_init, (1, H)), lb, ub, mp.ftol, mp.maxiter)
f, g, c, a = co.eval_fd(rows(np.arange(N), b.X[:N])); b.F[:N], b.Cc[:N], b.G[:N, :n], b.NZ[:N] = f, c, g, a.reshape(N, -1)[:, scatter]
while True:
    wf, wg = (w.copy() for w in b.advance())
    if len(wf) + len(wg) == 0: break
    if len(wf):
        f, c = co.eval(rows(wf, np.clip(b.X[wf], lb, ub))); b.F[wf], b.Cc[wf] = f, c
    if len(wg):
        _, g, _, a = co.eval_fd(rows(wg, np.clip(b.X[wg], lb, ub))); b.G[wg, :n] = g; b.NZ[wg] = a.reshape(len(wg), -1)[:, scatter]
lib.mpc_slsqp_set_lsq(None)
print(f"{len(rec)} QPs recorded; host core {1e3 * t_host[0] / len(rec):.3f} ms per QP (1 thread)", flush=True)
ctx = MpcContext(mp, K, 0)
reps = (K + len(rec) - 1) // len(rec)
up = lambda k: torch.as_tensor(np.tile(np.stack([r[k] for r in rec]), (reps, 1))[:K]).to(ctx.device)
args = [up(k) for k in range(6)]
from mpc_amd import _ffi
L, G, Cc, NZ, LO, HI = args
dev = ctx.device
mg = m + 2 * n
gen = torch.Generator(device="cpu").manual_seed(0)
s_in = (1e-2 * torch.randn(K, n, generator=gen, dtype=torch.float64)).to(dev)
R0 = torch.zeros(K, mg, dtype=torch.float64, device=dev)
V0 = (0.9 * G).clone()
index = torch.arange(K, dtype=torch.int32, device=dev)
s_out = torch.empty(K, n, dtype=torch.float64, device=dev); r_out = torch.empty(K, m, dtype=torch.float64, device=dev)
mode = torch.zeros(K, dtype=torch.int32, device=dev)
for fl in (0, 2, 1):
    flags = torch.full((K,), fl, dtype=torch.int32, device=dev)
    for rep in range(2):
        Lw, Rw, Vw = L.clone(), R0.clone(), V0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _ffi.check(ctx._ctx, ctx._lib.mpc_qp_step(ctx._ctx, K, K, index.data_ptr(), flags.data_ptr(), LO.data_ptr(), HI.data_ptr(),
                   s_in.data_ptr(), Lw.data_ptr(), G.data_ptr(), Cc.data_ptr(), NZ.data_ptr(), Rw.data_ptr(), Vw.data_ptr(),
                   s_out.data_ptr(), r_out.data_ptr(), mode.data_ptr(), ctx._stream()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"flags={fl} K={K}: {dt*1e3:.1f} ms -> {K/dt:.0f} QP steps/s; modes ok {int((mode == 1).sum())}", flush=True)
