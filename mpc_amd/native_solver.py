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
ABI_VERSION = 2
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
    L.mpc_slsqp_set_offload.restype = ip
    L.mpc_slsqp_set_offload.argtypes = [vp, ip]
    L.mpc_slsqp_advance_qp.restype = ip
    L.mpc_slsqp_advance_qp.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p]
    for name in ("lo", "hi", "s", "r"):
        fn = getattr(L, f"mpc_slsqp_{name}")
        fn.restype = dp
        fn.argtypes = [vp]
    for name in ("qp_mode", "qp_flags"):
        fn = getattr(L, f"mpc_slsqp_{name}")
        fn.restype = i32p
        fn.argtypes = [vp]
    for name in ("pack", "unpack"):
        fn = getattr(L, f"mpc_slsqp_{name}")
        fn.restype = ip
        fn.argtypes = [vp, ip, i32p, ip, i32p, ip, i32p, vp, vp]
    L.mpc_slsqp_core_seconds.restype = C.c_double
    L.mpc_slsqp_core_seconds.argtypes = [vp]
    if L.mpc_slsqp_abi_version() != ABI_VERSION:
        raise ImportError(f"libmpc_slsqp ABI {L.mpc_slsqp_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = L
    return L


class NativeBatch:
    """ctypes view of one ``mpc_slsqp_batch``: the rows as numpy arrays over the library's memory."""

    def __init__(self, horizon: int, capacity: int, threads: int = 0, offload_qp=False):
        """offload_qp: False (the whole method on the host), True (the QP is the caller's; an inconsistent linearisation
        ends the solve with MODE_NEEDS_HOST) or 2 (the caller solves Kraft's augmented problem as well, slsqp_core.h)."""
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
        self._wq = np.zeros(capacity, dtype=np.int32)
        self.offload_qp = bool(offload_qp)
        if self.offload_qp:
            self._lib.mpc_slsqp_set_offload(self._b, 2 if offload_qp == 2 else 1)
            self.LO = view(L.mpc_slsqp_lo, (capacity, self.n + 1))
            self.HI = view(L.mpc_slsqp_hi, (capacity, self.n + 1))
            self.S = view(L.mpc_slsqp_s, (capacity, self.n + 1))
            self.R = view(L.mpc_slsqp_r, (capacity, self.m + 2 * self.n + 2))
            self.QPMODE = view(L.mpc_slsqp_qp_mode, (capacity,))
            self.QPFLAGS = view(L.mpc_slsqp_qp_flags, (capacity,))

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

    def advance_qp(self):
        """One round with the QP offloaded: (want f and c, want grad f and dc/dx, want their QP solved)."""
        nf, ng, nq = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        rc = self._lib.mpc_slsqp_advance_qp(self._b, i32(self._wf), C.byref(nf), i32(self._wg), C.byref(ng), i32(self._wq),
                                            C.byref(nq))
        if rc < 0:
            raise RuntimeError(f"mpc_slsqp_advance_qp failed ({rc})")
        return self._wf[:nf.value], self._wg[:ng.value], self._wq[:nq.value]

    @property
    def core_seconds(self) -> float:
        return float(self._lib.mpc_slsqp_core_seconds(self._b))

    def close(self):
        if getattr(self, "_b", None) is not None and self._b.value:
            for name in ("X", "F", "Cc", "G", "NZ", "mode", "iterations", "LO", "HI", "S", "R", "QPMODE", "QPFLAGS"):
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


MODE_NEEDS_HOST = 104      # slsqp_core.h: MPC_SLSQP_MODE_NEEDS_HOST


class _DeviceQpLane:
    """One half of a DeviceQpSLSQP: its own batch in the C library, device state, staging buffers, context
    (workspaces) and stream, so that two lanes can be in flight at once."""

    def __init__(self, ctx: MpcContext, threads: int, capacity: int):
        self.ctx = ctx
        p = ctx.params
        self.cap = cap = int(capacity)
        self.batch = NativeBatch(p.horizon, cap, threads, offload_qp=True)
        self.lb = np.array([b[0] for b in p.bounds()])
        self.ub = np.array([b[1] for b in p.bounds()])
        dev, n, m = ctx.device, p.n, p.m
        z = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        self._x_dev = z(cap, n)
        self._out = EvalResult(z(cap), z(cap, n), z(cap, m), None, z(cap, ctx.nz_count))
        self._sub = {False: ScenarioBatch.empty(dev, cap, p.horizon, False), True: ScenarioBatch.empty(dev, cap, p.horizon, True)}
        # per-problem state that never leaves the device
        self._L = z(cap, (n + 1) * n // 2 + 1)
        self._G, self._C, self._NZ = z(cap, n), z(cap, m), z(cap, ctx.nz_count)
        self._R, self._V = z(cap, m + 2 * n), z(cap, n)
        # staging, one copy each way per round (mpc_slsqp_pack / mpc_slsqp_unpack define the layouts)
        pinned = lambda count, dt: torch.empty(count, dtype=dt).pin_memory()
        self._in_h, self._in_d = pinned(cap * 3 * n, torch.float64), z(cap * 3 * n)
        self._out_h, self._out_d = pinned(cap * (n + m + 1), torch.float64), z(cap * (n + m + 1))
        self._iin_h, self._iin_d = pinned(2 * cap, torch.int32), torch.empty(2 * cap, dtype=torch.int32, device=dev)
        self._iout_h, self._iout_d = pinned(cap, torch.int32), torch.empty(cap, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)

    def _round(self, sc, wf, wg, wq):
        """Serve one round's requests: one staged copy in, the evaluations and the QP step, one staged copy out."""
        ctx, b, o = self.ctx, self.batch, self._out
        n, m = b.n, b.m
        nf, ng, nq = len(wf), len(wg), len(wq)
        i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        lib = b._lib
        if lib.mpc_slsqp_pack(b._b, nf, i32(wf), ng, i32(wg), nq, i32(wq), self._in_h.data_ptr(), self._iin_h.data_ptr()):
            raise RuntimeError("mpc_slsqp_pack: bad arguments")
        t_in, i_in = (nf + ng + 3 * nq) * n, nf + ng + 2 * nq
        D, I, O = self._in_d, self._iin_d, self._out_d
        D[:t_in].copy_(self._in_h[:t_in], non_blocking=True)
        I[:i_in].copy_(self._iin_h[:i_in], non_blocking=True)
        cut = lambda buf, start, rows, cols: buf[start:start + rows * cols].view(rows, cols)
        xf, xg = cut(D, 0, nf, n), cut(D, nf * n, ng, n)
        q0 = (nf + ng) * n
        lo, hi, s_in = cut(D, q0, nq, n), cut(D, q0 + nq * n, nq, n), cut(D, q0 + 2 * nq * n, nq, n)
        jf, jg, jq, fl = I[:nf], I[nf:nf + ng], I[nf + ng:nf + ng + nq], I[nf + ng + nq:i_in]
        o_g = nf * m
        o_s = o_g + ng * n
        o_r = o_s + nq * n
        o_f = o_r + nq * m
        t_out = o_f + nf
        c_out, g_out, s_out, r_out, f_out = cut(O, 0, nf, m), cut(O, o_g, ng, n), cut(O, o_s, nq, n), cut(O, o_r, nq, m), O[o_f:t_out]
        subs = self._sub[sc.u_prev is not None]
        if nf:
            ctx.eval(ctx.gather(sc, jf, subs), xf, out=EvalResult(f_out, None, c_out, None))
            self._C.index_copy_(0, jf.long(), c_out)
        if ng:
            ctx.eval_fd_nz(ctx.gather(sc, jg, subs), xg, want_con=False,
                           out=EvalResult(o.cost[:ng], g_out, None, None, o.con_jac_nz[:ng]))
            rows = jg.long()
            self._G.index_copy_(0, rows, g_out)
            self._NZ.index_copy_(0, rows, o.con_jac_nz[:ng])
        if nq:
            from . import _ffi
            _ffi.check(ctx._ctx, ctx._lib.mpc_qp_step(
                ctx._ctx, nq, self.cap, jq.data_ptr(), fl.data_ptr(), lo.data_ptr(), hi.data_ptr(), s_in.data_ptr(),
                self._L.data_ptr(), self._G.data_ptr(), self._C.data_ptr(), self._NZ.data_ptr(), self._R.data_ptr(),
                self._V.data_ptr(), s_out.data_ptr(), r_out.data_ptr(), self._iout_d.data_ptr(), ctx._stream()))
            self._iout_h[:nq].copy_(self._iout_d[:nq], non_blocking=True)
        self._out_h[:t_out].copy_(O[:t_out], non_blocking=True)
        torch.cuda.current_stream(ctx.device).synchronize()
        if lib.mpc_slsqp_unpack(b._b, nf, i32(wf), ng, i32(wg), nq, i32(wq), self._out_h.data_ptr(), self._iout_h.data_ptr()):
            raise RuntimeError("mpc_slsqp_unpack: bad arguments")

    def solve(self, sc: ScenarioBatch, x0: np.ndarray) -> BatchSolveResult:
        """All of this lane's problems (sc.count <= capacity), on the lane's stream; status MODE_NEEDS_HOST marks the
        problems the caller has to solve again on the host."""
        ctx, b = self.ctx, self.batch
        p = ctx.params
        N, n = sc.count, p.n
        with torch.cuda.stream(self.stream):
            b.begin(x0, self.lb, self.ub, p.ftol, p.maxiter)
            t0 = time.perf_counter()
            o = self._out
            self._x_dev[:N].copy_(torch.from_numpy(b.X[:N]))
            ctx.eval_fd_nz(sc, self._x_dev[:N], out=EvalResult(o.cost[:N], o.grad[:N], o.con[:N], None, o.con_jac_nz[:N]))
            self._G[:N].copy_(o.grad[:N])
            self._C[:N].copy_(o.con[:N])
            self._NZ[:N].copy_(o.con_jac_nz[:N])
            b.F[:N] = o.cost[:N].cpu().numpy()
            b.Cc[:N] = o.con[:N].cpu().numpy()
            b.G[:N, :n] = o.grad[:N].cpu().numpy()
            t_serve = time.perf_counter() - t0
            n_eval, n_eval_fd, n_qp, rounds = 0, N, 0, 0
            while True:
                want_f, want_g, want_q = b.advance_qp()
                rounds += 1
                if len(want_f) + len(want_g) + len(want_q) == 0:
                    break
                t0 = time.perf_counter()
                self._round(sc, want_f, want_g, want_q)
                n_eval += len(want_f)
                n_eval_fd += len(want_g)
                n_qp += len(want_q)
                t_serve += time.perf_counter() - t0
        res = BatchSolveResult(x=b.X[:N].copy(), fun=b.F[:N].copy(), nit=b.iterations[:N].astype(int),
                               status=b.mode[:N].astype(int), rounds=rounds, n_eval=n_eval, n_eval_fd=n_eval_fd,
                               t_core=b.core_seconds, t_serve=t_serve)
        res.n_qp = n_qp
        return res

    def close(self):
        self.batch.close()


def active_lanes(count: int, lanes: int, lane_capacity: int, min_per_lane: int) -> int:
    """How many of a DeviceQpSLSQP's lanes a batch of `count` problems is split over: one per `min_per_lane`
    problems (more only add launches and host threads), but enough for every part to fit its lane."""
    k = max(1, min(lanes, count // max(1, min_per_lane)))
    return max(k, (count + lane_capacity - 1) // lane_capacity)


class DeviceQpSLSQP:
    """Batched SLSQP with the quadratic subproblems on the GPU.

    The O(n^2)..O(n^3) linear algebra of every major iteration - the BFGS update of the LDL' factor and the QP
    (``mpc_qp_step``: one workgroup per problem, arithmetic identical to the host core) - runs on the device,
    where the factor, the derivatives and the multipliers of every problem stay resident; the host threads
    keep the O(n + m) logic of the method (merit function, line search, convergence tests).  Per QP only the box,
    the last step and the new step with its multipliers cross PCIe, staged into one copy each way per round.
    ``lanes`` > 1 splits a batch into that many parts, each with its own stream, context and host threads: while one
    part's kernels run, the others' host logic and copies proceed.  Results equal the host drivers' bit for
    bit.  A problem whose linearisation turns out inconsistent (Kraft's augmented problem) is solved again by the
    host-only driver."""


    def __init__(self, ctx: MpcContext, threads: int = 0, capacity: int = None, lanes: int = 4, min_per_lane: int = 2048):
        if not ctx.nz_supported():
            raise ValueError("the driver takes dc/dx in its structured form: the context needs MPC_ALGO_PREFIX")
        if not 1 <= lanes <= 4:
            raise ValueError("lanes must be 1..4")
        self.ctx = ctx
        self.cap = cap = int(capacity or ctx.max_scenarios)
        self.threads = threads
        # below ~2,000 problems per lane another lane only adds launches and host threads (measured: 256 problems take
        # 0.31 s on four lanes, 0.08 s per 64 on one)
        self.min_per_lane = max(1, int(min_per_lane))
        lanes = max(1, min(lanes, cap // self.min_per_lane))
        lane_cap = (cap + lanes - 1) // lanes
        lane_threads = threads if lanes == 1 or threads <= 0 else max(1, (threads + lanes - 1) // lanes)
        self._ctxs = [ctx] + [MpcContext(ctx.params, lane_cap, ctx.device_index) for _ in range(lanes - 1)]
        self.lanes = [_DeviceQpLane(c, lane_threads, lane_cap) for c in self._ctxs]
        if lanes > 1:
            from concurrent.futures import ThreadPoolExecutor

            self._helper = ThreadPoolExecutor(max_workers=lanes - 1)
        self._host_fallback = None

    def solve(self, sc: ScenarioBatch, x0: np.ndarray) -> BatchSolveResult:
        ctx = self.ctx
        p = ctx.params
        N, n = sc.count, p.n
        if N > self.cap:
            raise ValueError(f"{N} scenarios > capacity {self.cap}")
        if N == 0:
            return BatchSolveResult(np.zeros((0, n)), np.zeros(0), np.zeros(0, int), np.zeros(0, int), 0, 0, 0)
        x0 = np.asarray(x0, dtype=np.float64).reshape(N, n)
        t_start = time.perf_counter()
        caller = torch.cuda.current_stream(ctx.device)
        for lane in self.lanes:
            lane.stream.wait_stream(caller)              # the scenarios were written on the caller's stream
        k = active_lanes(N, len(self.lanes), self.lanes[0].cap, self.min_per_lane)
        if k == 1:
            parts = [self.lanes[0].solve(sc, x0)]
        else:
            cuts = [(N * i) // k for i in range(k + 1)]
            others = [self._helper.submit(self.lanes[i].solve, sc.slice(cuts[i], cuts[i + 1]), x0[cuts[i]:cuts[i + 1]])
                      for i in range(1, k)]
            try:
                first = self.lanes[0].solve(sc.slice(0, cuts[1]), x0[:cuts[1]])
            finally:
                rest = [o.result() for o in others]
            parts = [first] + rest
        x = np.concatenate([r.x for r in parts])
        fun = np.concatenate([r.fun for r in parts])
        nit = np.concatenate([r.nit for r in parts])
        status = np.concatenate([r.status for r in parts])
        n_eval, n_eval_fd = sum(r.n_eval for r in parts), sum(r.n_eval_fd for r in parts)
        redo = np.nonzero(status == MODE_NEEDS_HOST)[0]
        if len(redo):                       # inconsistent linearisation somewhere along the way: host-only solve
            if self._host_fallback is None:
                self._host_fallback = NativeBatchedSLSQP(ctx, threads=self.threads, capacity=self.cap, groups=1)
            sub = ctx.gather(sc, torch.from_numpy(redo.astype(np.int32)).to(ctx.device),
                             ScenarioBatch.empty(ctx.device, len(redo), p.horizon, sc.u_prev is not None))
            r2 = self._host_fallback.solve(sub, x0[redo])
            x[redo], fun[redo], nit[redo], status[redo] = r2.x, r2.fun, r2.nit, r2.status
            n_eval += r2.n_eval
            n_eval_fd += r2.n_eval_fd
        res = BatchSolveResult(x=x, fun=fun, nit=nit, status=status, rounds=max(r.rounds for r in parts), n_eval=n_eval,
                               n_eval_fd=n_eval_fd, t_core=max(r.t_core for r in parts), t_serve=max(r.t_serve for r in parts))
        res.n_qp, res.t_qp, res.n_host_fallback = sum(r.n_qp for r in parts), 0.0, len(redo)
        res.t_wall = time.perf_counter() - t_start
        return res

    def close(self):
        for lane in self.lanes:
            lane.close()
        if getattr(self, "_helper", None) is not None:
            self._helper.shutdown()
            self._helper = None
        if self._host_fallback is not None:
            self._host_fallback.close()
            self._host_fallback = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
