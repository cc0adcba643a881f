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
    def __init__(self, ctx: MpcContext, threads: int = 0, capacity: int = None):
        if not ctx.nz_supported():
            raise ValueError("the native driver takes dc/dx in its structured form: the context needs MPC_ALGO_PREFIX")
        self.ctx = ctx
        p = ctx.params
        self.cap = int(capacity or ctx.max_scenarios)
        self.batch = NativeBatch(p.horizon, self.cap, threads)
        self.lb = np.array([b[0] for b in p.bounds()])
        self.ub = np.array([b[1] for b in p.bounds()])
        dev, n, m = ctx.device, p.n, p.m
        z = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        self._x_dev = z(self.cap, n)
        self._out = EvalResult(z(self.cap), z(self.cap, n), z(self.cap, m), None, z(self.cap, ctx.nz_count))
        self._sub = {False: ScenarioBatch.empty(dev, self.cap, p.horizon, False),
                     True: ScenarioBatch.empty(dev, self.cap, p.horizon, True)}

    def _serve(self, sc: ScenarioBatch, idx: np.ndarray, need_grad: bool):
        ctx, b, o = self.ctx, self.batch, self._out
        K, N = len(idx), sc.count
        full = K == N
        rows = slice(0, N) if full else idx
        self._x_dev[:K].copy_(torch.from_numpy(np.clip(b.X[rows], self.lb, self.ub)))       # _clip_x_for_func
        sub = sc if full else ctx.gather(sc, torch.from_numpy(idx).to(ctx.device), self._sub[sc.u_prev is not None])
        if need_grad:
            ctx.eval_fd_nz(sub, self._x_dev[:K], want_con=False,
                           out=EvalResult(o.cost[:K], o.grad[:K], None, None, o.con_jac_nz[:K]))
            b.G[rows, :b.n] = o.grad[:K].cpu().numpy()
            b.NZ[rows] = o.con_jac_nz[:K].cpu().numpy()
        else:
            ctx.eval(sub, self._x_dev[:K], out=EvalResult(o.cost[:K], None, o.con[:K], None))
            b.F[rows] = o.cost[:K].cpu().numpy()
            b.Cc[rows] = o.con[:K].cpu().numpy()

    def solve(self, sc: ScenarioBatch, x0: np.ndarray) -> BatchSolveResult:
        ctx, b = self.ctx, self.batch
        p = ctx.params
        N, n = sc.count, p.n
        if N > self.cap:
            raise ValueError(f"{N} scenarios > capacity {self.cap}")
        if N == 0:
            return BatchSolveResult(np.zeros((0, n)), np.zeros(0), np.zeros(0, int), np.zeros(0, int), 0, 0, 0)
        b.begin(np.asarray(x0, dtype=np.float64).reshape(N, n), self.lb, self.ub, p.ftol, p.maxiter)
        t_serve = 0.0
        t0 = time.perf_counter()
        # mode 0 on entry: f, c, grad f, dc/dx at the start point for everyone (_slsqp_py.py:419-425)
        o = self._out
        self._x_dev[:N].copy_(torch.from_numpy(b.X[:N]))
        ctx.eval_fd_nz(sc, self._x_dev[:N], out=EvalResult(o.cost[:N], o.grad[:N], o.con[:N], None, o.con_jac_nz[:N]))
        b.F[:N] = o.cost[:N].cpu().numpy()
        b.Cc[:N] = o.con[:N].cpu().numpy()
        b.G[:N, :n] = o.grad[:N].cpu().numpy()
        b.NZ[:N] = o.con_jac_nz[:N].cpu().numpy()
        t_serve += time.perf_counter() - t0
        n_eval, n_eval_fd, rounds = 0, N, 0
        while True:
            want_f, want_g = b.advance()
            rounds += 1
            if len(want_f) == 0 and len(want_g) == 0:
                break
            t0 = time.perf_counter()
            if len(want_f):
                self._serve(sc, want_f.copy(), False)
                n_eval += len(want_f)
            if len(want_g):
                self._serve(sc, want_g.copy(), True)
                n_eval_fd += len(want_g)
            t_serve += time.perf_counter() - t0
        return BatchSolveResult(x=b.X[:N].copy(), fun=b.F[:N].copy(), nit=b.iterations[:N].astype(int),
                                status=b.mode[:N].astype(int), rounds=rounds, n_eval=n_eval, n_eval_fd=n_eval_fd,
                                t_core=b.core_seconds, t_serve=t_serve)

    def close(self):
        self.batch.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
