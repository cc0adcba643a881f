"""Host-side state of one SLSQP instance and one reverse-communication step of scipy's compiled core.

Kept free of torch so that pool workers (mpc_amd/pool.py) start quickly and never touch the GPU."""
from __future__ import annotations

import numpy as np

try:  # private f2py entry point of scipy <= 1.15 (signature checked in tests/test_abi_load.py)
    from scipy.optimize._slsqp import slsqp as _slsqp_core
    _slsqp_import_error = None
except Exception as exc:  # pragma: no cover
    _slsqp_core = None
    _slsqp_import_error = exc

EXIT_MESSAGES = {
    -1: "Gradient evaluation required (g & a)", 0: "Optimization terminated successfully",
    1: "Function evaluation required (f & c)", 2: "More equality constraints than independent variables",
    3: "More than 3*n iterations in LSQ subproblem", 4: "Inequality constraints incompatible",
    5: "Singular matrix E in LSQ subproblem", 6: "Singular matrix C in LSQ subproblem",
    7: "Rank-deficient equality constraint subproblem HFTI", 8: "Positive directional derivative for linesearch",
    9: "Iteration limit reached",
}


def require_core():
    if _slsqp_core is None:
        raise ImportError(f"scipy.optimize._slsqp.slsqp is not importable: {_slsqp_import_error}")


class Problem:
    """The arrays `_minimize_slsqp` allocates for one problem (_slsqp_py.py:343-409)."""

    __slots__ = ("x", "fx", "c", "g", "a", "w", "jw", "mode", "majiter", "acc", "scal", "ints", "done", "m")

    def __init__(self, x0, n, m, acc, maxiter, a_view):
        n1 = n + 1
        meq = 0
        mineq = m - meq + n1 + n1
        len_w = ((3 * n1 + m) * (n1 + 1) + (n1 - meq + 1) * (mineq + 2) + 2 * mineq + (n1 + mineq) * (n1 - meq)
                 + 2 * meq + n1 + ((n + 1) * n) // 2 + 2 * m + 3 * n + 3 * n1 + 1)
        self.m = m
        self.x = x0                          # updated in place by the core
        self.fx = 0.0
        self.c = np.zeros(m)
        self.g = np.zeros(n1)
        self.a = a_view                      # (m, n+1) Fortran-ordered
        self.w = np.zeros(len_w)
        self.jw = np.zeros(mineq)
        self.mode = np.array(0, int)
        self.majiter = np.array(maxiter - 1, int)
        self.acc = np.array(acc, float)
        self.scal = [np.array(0, float) for _ in range(10)]  # alpha, f0, gs, h1..h4, t, t0, tol
        self.ints = [np.array(0, int) for _ in range(8)]     # iexact, incons, ireset, itermx, line, n1, n2, n3
        self.done = False

    def step(self, xl, xu) -> int:
        """One call of the core (_slsqp_py.py:429-433): returns the new mode (1: wants f and c,
        -1: wants grad f and dc/dx, anything else: finished)."""
        s, it = self.scal, self.ints
        _slsqp_core(self.m, 0, self.x, xl, xu, self.fx, self.c, self.g, self.a, self.acc, self.majiter, self.mode,
                    self.w, self.jw, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9],
                    it[0], it[1], it[2], it[3], it[4], it[5], it[6], it[7])
        md = int(self.mode)
        if md != 1 and md != -1:
            self.done = True
        return md


def nz_count(H: int) -> int:
    return 6 * H + 3 * H * (H + 1)


def nz_scatter_index(H: int) -> np.ndarray:
    """Flat positions, inside one problem's (n+1, m) C-ordered array (= the core's column-major
    a(la, n+1)), of the entries of the structured dc/dx (include/mpc_amd.h, mpc_eval_fd_nz), in the
    structured order: ``A.reshape(-1)[nz_scatter_index(H)] = nz`` expands one scenario."""
    m = 12 * (H + 1)
    idx = np.empty(nz_count(H), dtype=np.int64)
    for j in range(H):
        tri = j * H - (j * (j - 1)) // 2 - j - 1
        for a in range(6):
            col = 6 * j + a
            idx[col] = col * m + 12 * (j + 1) + 6 + a
            for k in range(j + 1, H + 1):
                idx[6 * H + 6 * (tri + k) + a] = col * m + 12 * k + a
    return idx


class SharedViews:
    """numpy views of the pool's shared-memory block.  Capacity `cap` scenarios.

    X    [cap,n]    iterates (the workers' cores update their rows in place, the server reads them)
    SF   [cap]      staged f          worker w reads its k staged rows at [first_w, first_w + k)
    SC   [cap,m]    staged c
    SG   [cap,n]    staged grad f
    SJ   [cap,jw]   staged dc/dx: structured (jw = nz_count(H), include/mpc_amd.h) or dense column-major
                    m x n (jw = n*m) when the context cannot produce the structured form
    IDXF [cap] i4   scenarios of worker w that want (f, c), listed at [first_w, ...)
    IDXG [cap] i4   scenarios that want (grad f, dc/dx)
    """

    FIELDS = ("X", "SF", "SC", "SG", "SJ")

    @staticmethod
    def nbytes(cap, n, m, jw):
        return 8 * cap * (n + 1 + m + n + jw) + 2 * 4 * cap

    def __init__(self, buf, cap, n, m, jw):
        off = 0

        def take(shape, dtype=np.float64):
            nonlocal off
            a = np.ndarray(shape, dtype=dtype, buffer=buf, offset=off)
            off += a.nbytes
            return a

        self.X = take((cap, n))
        self.SF = take((cap,))
        self.SC = take((cap, m))
        self.SG = take((cap, n))
        self.SJ = take((cap, jw))
        self.IDXF = take((cap,), np.int32)
        self.IDXG = take((cap,), np.int32)


def worker_main(conn, shm_name, cap, n, m, structured):
    """Pool worker: owns the SLSQP cores of a contiguous range of scenarios.  Never touches the GPU."""
    import time
    from multiprocessing import shared_memory

    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        H = n // 6
        v = SharedViews(shm.buf, cap, n, m, nz_count(H) if structured else n * m)
        scatter = nz_scatter_index(H) if structured else None
        conn.send(("ready",))
        while True:
            msg = conn.recv()
            if msg[0] == "stop":
                break
            _, first, cnt, acc, maxiter, xl, xu = msg
            try:
                _solve_range(conn, v, scatter, first, cnt, n, m, acc, maxiter, xl, xu, time)
            except Exception as exc:  # noqa: BLE001 - reported to the server
                conn.send(("error", repr(exc)))
        del v
    finally:
        shm.close()


def _solve_range(conn, v, scatter, first, cnt, n, m, acc, maxiter, xl, xu, time):
    A = np.zeros((cnt, n + 1, m))
    Aflat = A.reshape(cnt, -1)
    probs = [Problem(v.X[first + i], n, m, acc, maxiter, A[i].T) for i in range(cnt)]
    want_f, want_g = [], list(range(cnt))
    initial = True
    t_core = 0.0
    rounds = 0
    while True:
        tag = conn.recv()
        if tag[0] != "data":
            raise RuntimeError(f"unexpected message {tag!r}")
        for j, i in enumerate(want_f):
            p = probs[i]
            p.fx = float(v.SF[first + j])
            p.c[:] = v.SC[first + j]
        for j, i in enumerate(want_g):
            p = probs[i]
            p.g[:n] = v.SG[first + j]
            p.g[n] = 0.0
            if scatter is not None:
                Aflat[i, scatter] = v.SJ[first + j]          # every other entry stays zero
            else:
                A[i, :n, :] = v.SJ[first + j].reshape(n, m)
            if initial:                      # mode 0 on entry: f, c, g, a at the start point
                p.fx = float(v.SF[first + j])
                p.c[:] = v.SC[first + j]
        initial = False
        t0 = time.perf_counter()
        want_f, want_g = [], []
        for i, p in enumerate(probs):
            if p.done:
                continue
            md = p.step(xl, xu)
            if md == 1:
                want_f.append(i)
            elif md == -1:
                want_g.append(i)
        t_core += time.perf_counter() - t0
        rounds += 1
        if not want_f and not want_g:
            conn.send(("done", np.array([p.fx for p in probs]), np.array([int(p.majiter) for p in probs]),
                       np.array([int(p.mode) for p in probs]), t_core, rounds))
            return
        v.IDXF[first:first + len(want_f)] = np.asarray(want_f, np.int32) + first
        v.IDXG[first:first + len(want_g)] = np.asarray(want_g, np.int32) + first
        conn.send(("req", len(want_f), len(want_g)))
