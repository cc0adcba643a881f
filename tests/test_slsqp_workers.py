"""CPU tests of the host half of a batched solve (mpc_amd/slsqp_state.py): the reverse-communication
step against scipy's own driver, and the pool's worker protocol with the C oracle standing in for
the GPU server (test infrastructure only: the product serves these requests from libmpc_amd)."""
import multiprocessing as mp_
from multiprocessing import connection, shared_memory

import numpy as np
import pytest
from scipy.optimize import minimize

from helpers import host_scenarios, params_for
from mpc_amd.sharding import shard_range
from mpc_amd.slsqp_state import Problem, SharedViews, nz_count, nz_scatter_index, worker_main


def _row(hs, i, x):
    d = hs.as_oracle_dict(i, i + 1)
    d["x"] = np.asarray(x, float).reshape(1, -1)
    return d


def _lockstep_one(co, hs, i, mp, lb, ub):
    n, m = mp.n, mp.m
    x = np.clip(hs.x[i].copy(), lb, ub)
    A = np.zeros((n + 1, m))
    p = Problem(x, n, m, mp.ftol, mp.maxiter, A.T)
    f, g, c, a = co.eval_fd(_row(hs, i, x))
    p.fx, p.c[:], p.g[:n], A[:n] = float(f[0]), c[0], g[0], a[0]
    while not p.done:
        md = p.step(lb, ub)
        xc = np.clip(p.x, lb, ub)
        if md == 1:
            f, c = co.eval(_row(hs, i, xc))
            p.fx, p.c[:] = float(f[0]), c[0]
        elif md == -1:
            _, g, _, a = co.eval_fd(_row(hs, i, xc))
            p.g[:n], A[:n] = g[0], a[0]
    return p


def test_step_reproduces_scipy_minimize():
    """Problem.step driven by hand == scipy.optimize.minimize(method='SLSQP') on the same callables."""
    mp = params_for(3)
    co, hs = host_scenarios(mp, 4, motion="angular", cfg_id=5)
    lb = np.array([b[0] for b in mp.bounds()])
    ub = np.array([b[1] for b in mp.bounds()])
    for i in range(4):
        p = _lockstep_one(co, hs, i, mp, lb, ub)
        res = minimize(fun=lambda z: float(co.eval(_row(hs, i, z))[0][0]), x0=hs.x[i],
                       jac=lambda z: co.eval_fd(_row(hs, i, z), want_jac=False)[1][0], method="SLSQP", bounds=mp.bounds(),
                       constraints=[{"type": "ineq", "fun": lambda z: co.eval(_row(hs, i, z))[1][0],
                                     "jac": lambda z: co.eval_fd(_row(hs, i, z))[3][0].T}],
                       options={"ftol": mp.ftol, "maxiter": mp.maxiter})
        np.testing.assert_array_equal(p.x, res.x)
        assert int(p.mode) == res.status and int(p.majiter) == res.nit
        assert p.fx == res.fun


@pytest.mark.parametrize("structured", [False, True])
def test_worker_protocol_equals_in_process_loop(structured):
    """Workers fed dense dc/dx, or its structured form (include/mpc_amd.h) which they expand themselves."""
    mp = params_for(3)
    N, W = 7, 3                     # 3 + 2 + 2 scenarios
    co, hs = host_scenarios(mp, N, motion="linear", cfg_id=6)
    n, m = mp.n, mp.m
    lb = np.array([b[0] for b in mp.bounds()])
    ub = np.array([b[1] for b in mp.bounds()])
    want = [_lockstep_one(co, hs, i, mp, lb, ub) for i in range(N)]

    jw = nz_count(mp.horizon) if structured else n * m
    pack = (lambda a: a.reshape(-1)[nz_scatter_index(mp.horizon)]) if structured else (lambda a: a.reshape(-1))
    shm = shared_memory.SharedMemory(create=True, size=SharedViews.nbytes(N, n, m, jw))
    spawn = mp_.get_context("spawn")
    conns, procs = [], []
    try:
        v = SharedViews(shm.buf, N, n, m, jw)
        for _ in range(W):
            a, b = spawn.Pipe()
            pr = spawn.Process(target=worker_main, args=(b, shm.name, N, n, m, structured), daemon=True)
            pr.start()
            b.close()
            conns.append(a)
            procs.append(pr)
        assert all(c.recv()[0] == "ready" for c in conns)
        v.X[:] = np.clip(hs.x, lb, ub)
        busy = {}
        for r, c in enumerate(conns):
            first, cnt = shard_range(N, W, r)
            c.send(("solve", first, cnt, mp.ftol, mp.maxiter, lb, ub))
            busy[c] = (first, cnt)
        d = hs.as_oracle_dict()
        d["x"] = v.X.copy()
        v.SF[:], v.SG[:], v.SC[:], jac0 = co.eval_fd(d)
        v.SJ[:] = np.stack([pack(a) for a in jac0])
        for c in conns:
            c.send(("data",))
        fun, status = np.zeros(N), np.zeros(N, int)
        while busy:
            for c in connection.wait(list(busy)):
                msg = c.recv()
                first, cnt = busy[c]
                assert msg[0] in ("req", "done"), msg
                if msg[0] == "done":
                    fun[first:first + cnt], status[first:first + cnt] = msg[1], msg[3]
                    del busy[c]
                    continue
                kf, kg = msg[1], msg[2]
                for j, s in enumerate(v.IDXF[first:first + kf]):
                    f, cc = co.eval(_row(hs, int(s), np.clip(v.X[s], lb, ub)))
                    v.SF[first + j], v.SC[first + j] = f[0], cc[0]
                for j, s in enumerate(v.IDXG[first:first + kg]):
                    _, g, _, a = co.eval_fd(_row(hs, int(s), np.clip(v.X[s], lb, ub)))
                    v.SG[first + j], v.SJ[first + j] = g[0], pack(a[0])
                c.send(("data",))
        X = v.X.copy()
        del v
    finally:
        for c in conns:
            try:
                c.send(("stop",))
            except Exception:
                pass
        for pr in procs:
            pr.join(timeout=10)
        shm.close()
        shm.unlink()
    np.testing.assert_array_equal(X, np.array([p.x for p in want]))
    np.testing.assert_array_equal(status, np.array([int(p.mode) for p in want]))
    np.testing.assert_array_equal(fun, np.array([p.fx for p in want]))
