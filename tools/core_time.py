"""Dev tool (CPU only): time the native SLSQP core on H=20 problems, the C oracle serving the evaluations."""
import os
import sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import host_scenarios, params_for
from mpc_amd.native_solver import NativeBatch
from mpc_amd.slsqp_state import nz_scatter_index
H, N = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 32
mp = params_for(H)
co, hs = host_scenarios(mp, N, motion="angular", cfg_id=3, goal_jitter=0.0)
n = mp.n
lb = np.array([b[0] for b in mp.bounds()]); ub = np.array([b[1] for b in mp.bounds()])
scatter = nz_scatter_index(H)
def rows(idx, X):
    d = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}; d["x"] = np.ascontiguousarray(X); return d
b = NativeBatch(H, N, threads=1)
b.begin(np.tile(hs.u_init, (1, H)), lb, ub, mp.ftol, mp.maxiter)
f, g, c, a = co.eval_fd(rows(np.arange(N), b.X[:N]))
b.F[:N], b.Cc[:N], b.G[:N, :n], b.NZ[:N] = f, c, g, a.reshape(N, -1)[:, scatter]
while True:
    wf, wg = (w.copy() for w in b.advance())
    if len(wf) + len(wg) == 0: break
    if len(wf):
        f, c = co.eval(rows(wf, np.clip(b.X[wf], lb, ub))); b.F[wf], b.Cc[wf] = f, c
    if len(wg):
        _, g, _, a = co.eval_fd(rows(wg, np.clip(b.X[wg], lb, ub))); b.G[wg, :n] = g; b.NZ[wg] = a.reshape(len(wg), -1)[:, scatter]
print(f"core seconds {b.core_seconds:.3f} for {N} solves; checksum {b.X[:N].sum():.17g} nit {b.iterations[:N].sum()}")
