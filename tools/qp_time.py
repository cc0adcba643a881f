"""Dev tool: throughput of mpc_qp_solve on QPs recorded from host solves (H=20), tiled to a large batch."""
import os, sys; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import ctypes, time
import numpy as np, torch
from helpers import host_scenarios, params_for
from mpc_amd import native_solver
from mpc_amd.batch import MpcContext
from mpc_amd.native_solver import NativeBatch
from mpc_amd.slsqp_state import nz_scatter_index

H, N = 20, 6
K = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 4096
FIRST_ONLY = "--first" in sys.argv      # only the first QP of every solve (identity Hessian, many active bounds)
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
if FIRST_ONLY:
    rec = [r for r in rec if np.count_nonzero(r[0][:-1]) == n]
print(f"{len(rec)} QPs recorded; host core {1e3 * t_host[0] / len(rec):.3f} ms per QP (1 thread)", flush=True)
ctx = MpcContext(mp, K, 0)
reps = (K + len(rec) - 1) // len(rec)
up = lambda k: torch.as_tensor(np.tile(np.stack([r[k] for r in rec]), (reps, 1))[:K]).to(ctx.device)
args = [up(k) for k in range(6)]
s, y, mode = ctx.qp_solve(*args); torch.cuda.synchronize()
want = np.tile(np.stack([r[6] for r in rec]), (reps, 1))[:K]
print("bit-identical:", np.array_equal(s.cpu().numpy(), want), "modes ok:", int((mode == 1).sum()), "of", K, flush=True)
for _ in range(2):
    t0 = time.perf_counter(); ctx.qp_solve(*args); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"K={K}: {dt*1e3:.1f} ms -> {K/dt:.0f} QPs/s", flush=True)
