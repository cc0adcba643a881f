"""Batched SLSQP on the native host driver (libmpc_slsqp.so, include/mpc_slsqp.h).

Same job as :class:`mpc_amd.solver.BatchedSLSQP` / :class:`mpc_amd.pool.SolverPool`, but the
per-problem SLSQP state and the core itself live in C: one ``mpc_slsqp_advance`` call per round
advances every unfinished problem on the library's own host threads (no GIL, no worker processes,
no dependency on scipy's private ``_slsqp`` module), and this process serves each round's requests
with one ``mpc_eval`` and one ``mpc_eval_fd_nz`` launch.  The C core reproduces scipy's compiled
core bit for bit (tests/test_native_slsqp.py), so results equal the other two drivers' row for row.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch

from .batch import EvalResult, MpcContext, ScenarioBatch
from .solver import BatchSolveResult

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPC_SLSQP_LIB") or os.path.join(_HERE, "libmpc_slsqp.so")
ABI_VERSION = 1
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m mpc_amd.build`")
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
    i32p = C.POINTER(C.c_int32)
    L.mpc_slsqp_abi_version.restype = ip
    L.mpc_slsqp_create.restype = ip
    L.mpc_slsqp_create.argtypes = [ip, ip, ip, C.POINTER(vp)]
    L.mpc_slsqp_destroy.restype = None
    L.mpc_slsqp_destroy.argtypes = [vp]
    L.mpc_slsqp_begin.restype = ip
    L.mpc_slsqp_begin.argtypes = [vp, ip, dp, dp, dp, C.c_double, ip]
    for name in ("x", "f", "c", "g", "nz"):
        fn = getattr(L, f"mpc_slsqp_{name}")
        fn.restype = dp
        fn.argtypes = [vp]
    for name in ("mode", "iterations"):
        fn = getattr(L, f"mpc_slsqp_{name}")
        fn.restype = i32p
        fn.argtypes = [vp]
    L.mpc_slsqp_advance.restype = ip
    L.mpc_slsqp_advance.argtypes = [vp, i32p, i32p, i32p, i32p]
    L.mpc_slsqp_core_seconds.restype = C.c_double
    L.mpc_slsqp_core_seconds.argtypes = [vp]
    if L.mpc_slsqp_abi_version() != ABI_VERSION:
        raise ImportError(f"libmpc_slsqp ABI {L.mpc_slsqp_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = L
    return L


class NativeBatch:
    """ctypes view of one ``mpc_slsqp_batch``: the rows as numpy arrays over the library's memory."""

    def __init__(self, horizon: int, capacity: int, threads: int = 0):
        self._lib = load()
        self._b = C.c_void_p()
        rc = self._lib.mpc_slsqp_create(horizon, capacity, threads, C.byref(self._b))
        if rc:
            raise MemoryError(f"mpc_slsqp_create failed ({rc})") if rc == -2 else ValueError(f"mpc_slsqp_create: bad arguments ({rc})")
        H = horizon
        self.H, self.cap = H, capacity
        self.n, self.m = 6 * H, 12 * (H + 1)
        self.nzc = 6 * H + 3 * H * (H + 1)
        view = lambda fn, shape: np.ctypeslib.as_array(fn(self._b), shape=shape)
        L = self._lib
        self.X = view(L.mpc_slsqp_x, (capacity, self.n))
        self.F = view(L.mpc_slsqp_f, (capacity,))
        self.Cc = view(L.mpc_slsqp_c, (capacity, self.m))
        self.G = view(L.mpc_slsqp_g, (capacity, self.n + 1))
        self.NZ = view(L.mpc_slsqp_nz, (capacity, self.nzc))
        self.mode = view(L.mpc_slsqp_mode, (capacity,))
        self.iterations = view(L.mpc_slsqp_iterations, (capacity,))
        self._wf = np.zeros(capacity, dtype=np.int32)
        self._wg = np.zeros(capacity, dtype=np.int32)

    def begin(self, x0: np.ndarray, lb: np.ndarray, ub: np.ndarray, acc: float, maxiter: int):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        count = x0.shape[0]
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        lb, ub = np.ascontiguousarray(lb, dtype=np.float64), np.ascontiguousarray(ub, dtype=np.float64)
        rc = self._lib.mpc_slsqp_begin(self._b, count, dp(x0), dp(lb), dp(ub), float(acc), int(maxiter))
        if rc:
            raise ValueError(f"mpc_slsqp_begin: bad arguments ({rc})")
        self.count = count

    def advance(self):
        """One round.  Returns (indices wanting f and c, indices wanting grad f and dc/dx)."""
        nf, ng = C.c_int32(0), C.c_int32(0)
        i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        rc = self._lib.mpc_slsqp_advance(self._b, i32(self._wf), C.byref(nf), i32(self._wg), C.byref(ng))
        if rc < 0:
            raise RuntimeError(f"mpc_slsqp_advance failed ({rc})")
        return self._wf[:nf.value], self._wg[:ng.value]

    @property
    def core_seconds(self) -> float:
        return float(self._lib.mpc_slsqp_core_seconds(self._b))

    def close(self):
        if getattr(self, "_b", None) is not None and self._b.value:
            for name in ("X", "F", "Cc", "G", "NZ", "mode", "iterations"):
                setattr(self, name, None)
            self._lib.mpc_slsqp_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class NativeBatchedSLSQP:
    """`groups` = 2 splits a batch into two halves that alternate: while the host threads advance one
    half inside the C library (the GIL is released), this thread serves the other half's evaluations
    from the GPU, so device calls and copies hide behind the cores."""

    def __init__(self, ctx: MpcContext, threads: int = 0, capacity: int = None, groups: int = 2):
        if not ctx.nz_supported():
            raise ValueError("the native driver takes dc/dx in its structured form: the context needs MPC_ALGO_PREFIX")
        if groups not in (1, 2):
            raise ValueError("groups must be 1 or 2")
        self.ctx = ctx
        p = ctx.params
        self.cap = int(capacity or ctx.max_scenarios)
        self.groups = groups
        self._gcap = (self.cap + groups - 1) // groups
        self.batches = [NativeBatch(p.horizon, self._gcap, threads) for _ in range(groups)]
        self.lb = np.array([b[0] for b in p.bounds()])
        self.ub = np.array([b[1] for b in p.bounds()])
        dev, n, m = ctx.device, p.n, p.m
        z = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        self._x_dev = z(self.cap, n)
        self._out = EvalResult(z(self.cap), z(self.cap, n), z(self.cap, m), None, z(self.cap, ctx.nz_count))
        self._sub = {False: ScenarioBatch.empty(dev, self.cap, p.horizon, False),
                     True: ScenarioBatch.empty(dev, self.cap, p.horizon, True)}
        from concurrent.futures import ThreadPoolExecutor

        self._helper = ThreadPoolExecutor(max_workers=1)

    def _serve(self, sc: ScenarioBatch, b: NativeBatch, first: int, idx: np.ndarray, need_grad: bool):
        """Evaluate rows `idx` of group `b` (scenarios first + idx of the batch) in one launch."""
        ctx, o = self.ctx, self._out
        K = len(idx)
        self._x_dev[:K].copy_(torch.from_numpy(np.clip(b.X[idx], self.lb, self.ub)))       # _clip_x_for_func
        gidx = torch.from_numpy((idx + first).astype(np.int32)).to(ctx.device)
        sub = ctx.gather(sc, gidx, self._sub[sc.u_prev is not None])
        if need_grad:
            ctx.eval_fd_nz(sub, self._x_dev[:K], want_con=False,
                           out=EvalResult(o.cost[:K], o.grad[:K], None, None, o.con_jac_nz[:K]))
            b.G[idx, :b.n] = o.grad[:K].cpu().numpy()
            b.NZ[idx] = o.con_jac_nz[:K].cpu().numpy()
        else:
            ctx.eval(sub, self._x_dev[:K], out=EvalResult(o.cost[:K], None, o.con[:K], None))
            b.F[idx] = o.cost[:K].cpu().numpy()
            b.Cc[idx] = o.con[:K].cpu().numpy()

    def solve(self, sc: ScenarioBatch, x0: np.ndarray) -> BatchSolveResult:
        ctx = self.ctx
        p = ctx.params
        N, n = sc.count, p.n
        if N > self.cap:
            raise ValueError(f"{N} scenarios > capacity {self.cap}")
        if N == 0:
            return BatchSolveResult(np.zeros((0, n)), np.zeros(0), np.zeros(0, int), np.zeros(0, int), 0, 0, 0)
        x0 = np.asarray(x0, dtype=np.float64).reshape(N, n)
        G = self.groups if N >= 2 * self.groups else 1
        per = (N + G - 1) // G
        spans = [(g * per, min(N, (g + 1) * per) - g * per) for g in range(G)]
        bs = self.batches[:G]
        for b, (first, cnt) in zip(bs, spans):
            b.begin(x0[first:first + cnt], self.lb, self.ub, p.ftol, p.maxiter)
        t_serve = 0.0
        t0 = time.perf_counter()
        # mode 0 on entry: f, c, grad f, dc/dx at the start point for everyone (_slsqp_py.py:419-425)
        o = self._out
        for b, (first, cnt) in zip(bs, spans):
            self._x_dev[first:first + cnt].copy_(torch.from_numpy(b.X[:cnt]))
        ctx.eval_fd_nz(sc, self._x_dev[:N], out=EvalResult(o.cost[:N], o.grad[:N], o.con[:N], None, o.con_jac_nz[:N]))
        cost, con, grad, nz = (t[:N].cpu().numpy() for t in (o.cost, o.con, o.grad, o.con_jac_nz))
        for b, (first, cnt) in zip(bs, spans):
            b.F[:cnt], b.Cc[:cnt], b.G[:cnt, :n], b.NZ[:cnt] = (a[first:first + cnt] for a in (cost, con, grad, nz))
        t_serve += time.perf_counter() - t0
        n_eval, n_eval_fd, rounds = 0, N, 0
        requests = [None] * G          # per group: (want_f, want_g) still to be served, or None
        alive = [True] * G

        def advance(g):
            want_f, want_g = bs[g].advance()
            requests[g] = (want_f.copy(), want_g.copy())
            if len(want_f) + len(want_g) == 0:
                alive[g] = False

        def serve(g):
            nonlocal n_eval, n_eval_fd, t_serve
            req, requests[g] = requests[g], None
            if req is None:
                return
            t1 = time.perf_counter()
            first = spans[g][0]
            if len(req[0]):
                self._serve(sc, bs[g], first, req[0], False)
                n_eval += len(req[0])
            if len(req[1]):
                self._serve(sc, bs[g], first, req[1], True)
                n_eval_fd += len(req[1])
            t_serve += time.perf_counter() - t1

        # every advance runs on ONE helper thread, so that the library keeps one team of host threads
        run = self._helper.submit
        run(advance, 0).result()
        rounds += 1
        cur = 0
        while any(alive):
            other = (cur + 1) % G
            pending = None
            if G > 1 and alive[other] and requests[other] is None:
                pending = run(advance, other)  # the cores of the other half run while this half is served
            serve(cur)
            if pending is not None:
                pending.result()
                rounds += 1
            if G == 1 or not alive[other]:
                if alive[cur]:
                    run(advance, cur).result()
                    rounds += 1
            else:
                cur = other
        x = np.concatenate([b.X[:cnt] for b, (_, cnt) in zip(bs, spans)])
        cat = lambda name: np.concatenate([getattr(b, name)[:cnt] for b, (_, cnt) in zip(bs, spans)])
        return BatchSolveResult(x=x.copy(), fun=cat("F").copy(), nit=cat("iterations").astype(int),
                                status=cat("mode").astype(int), rounds=rounds, n_eval=n_eval, n_eval_fd=n_eval_fd,
                                t_core=sum(b.core_seconds for b in bs), t_serve=t_serve)

    def close(self):
        self._helper.shutdown(wait=True)
        for b in self.batches:
            b.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
