"""Host-side batch interface: torch tensors as device buffers, libmpc_amd.so for every number.

Torch is plumbing here (device memory, streams); nothing is computed with torch ops and there is
no fallback when the library or the GPU is missing."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _ffi
from .params import MpcParams


def _dev_index(device) -> int:
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError(f"mpc_amd runs on a HIP device only, got {d}")
    return d.index if d.index is not None else torch.cuda.current_device()


@dataclass
class ScenarioBatch:
    """Per-scenario inputs that ``compute_control`` assembles before calling ``minimize``
    (ur_mpc.py:148-157), N scenarios, each row contiguous, fp64 on the device."""

    state0: torch.Tensor                      # [N,18] = [q, dq, ee_pos, ee_rpy]  (ur20_dynamics.py:49-56)
    target: torch.Tensor                      # [N,6]  = [pos, rpy]
    base_traj: torch.Tensor                   # [N,H,12] = [position, orientation, lin vel, ang vel] per step
    Jc: torch.Tensor                          # [N,36] = _J_cached, row-major (ur_mpc.py:157)
    u_prev: Optional[torch.Tensor] = None     # [N,6]  previous_control
    has_prev: Optional[torch.Tensor] = None   # [N] uint8: does previous_control exist yet (ur_mpc.py:313)

    @property
    def count(self) -> int:
        return int(self.state0.shape[0])

    @classmethod
    def from_numpy(cls, device, state0, target, base_traj, Jc, u_prev=None, has_prev=None) -> "ScenarioBatch":
        dev = torch.device("cuda", _dev_index(device))

        def up(a, dtype=torch.float64):
            return None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)

        N = np.asarray(state0).reshape(-1, 18).shape[0]
        return cls(up(np.asarray(state0).reshape(N, 18)), up(np.asarray(target).reshape(N, 6)),
                   up(np.asarray(base_traj).reshape(N, -1, 12)), up(np.asarray(Jc).reshape(N, 36)),
                   up(None if u_prev is None else np.asarray(u_prev).reshape(N, 6)),
                   up(None if has_prev is None else np.asarray(has_prev).reshape(N), torch.uint8))

    @classmethod
    def empty(cls, device, count: int, horizon: int, with_prev: bool) -> "ScenarioBatch":
        dev = torch.device("cuda", _dev_index(device))
        z = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        return cls(z(count, 18), z(count, 6), z(count, horizon, 12), z(count, 36),
                   z(count, 6) if with_prev else None,
                   torch.zeros(count, dtype=torch.uint8, device=dev) if with_prev else None)

    def slice(self, lo: int, hi: int) -> "ScenarioBatch":
        f = lambda t: None if t is None else t[lo:hi]
        return ScenarioBatch(f(self.state0), f(self.target), f(self.base_traj), f(self.Jc), f(self.u_prev),
                             f(self.has_prev))


@dataclass
class EvalResult:
    cost: torch.Tensor                 # [N]
    grad: Optional[torch.Tensor]       # [N,n]
    con: Optional[torch.Tensor]        # [N,m]
    con_jac: Optional[torch.Tensor]    # [N,n,m]: per scenario column-major m x n
    con_jac_nz: Optional[torch.Tensor] = None   # [N, 6H+3H(H+1)]: structured form (eval_fd_nz)

    def jac_for_scipy(self, i: int) -> np.ndarray:
        """(m, n) Fortran-ordered array of scenario i, the layout SLSQP's core consumes."""
        return self.con_jac[i].cpu().numpy().T


class MpcContext:
    """One libmpc_amd context: fixed horizon, parameters and device."""

    def __init__(self, params: MpcParams, max_scenarios: int, device=0):
        self._lib = _ffi.load()
        self.params = params
        self.device_index = _dev_index(device)
        self.device = torch.device("cuda", self.device_index)
        self.max_scenarios = int(max_scenarios)
        self.H, self.n, self.m = params.horizon, params.n, params.m
        self._ctx = C.c_void_p()
        _ffi.check(None, self._lib.mpc_ctx_create(self.device_index, params.horizon, self.max_scenarios,
                                                  C.byref(self._ctx)))
        self.set_params(params)

    def set_params(self, params: MpcParams):
        if params.horizon != self.H:
            raise ValueError("horizon is fixed per context")
        cp = params.to_c()
        decay = params.decay_table()
        _ffi.check(self._ctx, self._lib.mpc_set_params(self._ctx, C.byref(cp),
                                                       decay.ctypes.data_as(C.POINTER(C.c_double)), len(decay)))
        self.params = params

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.mpc_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- argument plumbing ---------------------------------------------------------------
    def _chk(self, t: Optional[torch.Tensor], shape, name, dtype=torch.float64, optional=False):
        if t is None:
            if optional:
                return None
            raise ValueError(f"{name} is required")
        if not (t.is_cuda and t.device.index == self.device_index):
            raise ValueError(f"{name} must live on {self.device}")
        if t.dtype != dtype or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous {dtype}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
        return t.data_ptr()

    def _scen_c(self, sc: ScenarioBatch, limit: bool = True) -> _ffi.MpcScenariosC:
        N = sc.count
        if limit and N > self.max_scenarios:
            raise ValueError(f"{N} scenarios > max_scenarios {self.max_scenarios}")
        c = _ffi.MpcScenariosC()
        c.count = N
        c.state0 = self._chk(sc.state0, (N, 18), "state0")
        c.target = self._chk(sc.target, (N, 6), "target")
        c.base_traj = self._chk(sc.base_traj, (N, self.H, 12), "base_traj")
        c.Jc = self._chk(sc.Jc, (N, 36), "Jc")
        c.u_prev = self._chk(sc.u_prev, (N, 6), "u_prev", optional=True)
        c.has_prev = self._chk(sc.has_prev, (N,), "has_prev", dtype=torch.uint8, optional=True)
        return c

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float64, device=self.device)

    # ---- the hot path ----------------------------------------------------------------------
    def eval(self, sc: ScenarioBatch, x: torch.Tensor, want_con: bool = True, out: Optional[EvalResult] = None):
        """f(x), c(x): ``_cost_function`` (ur_mpc.py:665-692) and ``dynamics_constraint`` (ur_mpc.py:388-411)."""
        N = sc.count
        scc = self._scen_c(sc)
        xp = self._chk(x, (N, self.n), "x")
        cost = out.cost if out is not None else self._new(N)
        con = (out.con if out is not None else self._new(N, self.m)) if want_con else None
        self._chk(cost, (N,), "cost")
        cp = self._chk(con, (N, self.m), "con", optional=True)
        _ffi.check(self._ctx, self._lib.mpc_eval(self._ctx, C.byref(scc), xp, cost.data_ptr(), cp, self._stream()))
        return EvalResult(cost, None, con, None)

    def eval_fd(self, sc: ScenarioBatch, x: torch.Tensor, want_grad: bool = True, want_con: bool = True,
                want_jac: bool = True, out: Optional[EvalResult] = None) -> EvalResult:
        """f, c and scipy's forward-difference grad f and dc/dx (one "eval" per scenario)."""
        N = sc.count
        scc = self._scen_c(sc)
        xp = self._chk(x, (N, self.n), "x")
        cost = out.cost if out is not None else self._new(N)
        grad = (out.grad if out is not None and out.grad is not None else self._new(N, self.n)) if want_grad else None
        con = (out.con if out is not None and out.con is not None else self._new(N, self.m)) if want_con else None
        jac = (out.con_jac if out is not None and out.con_jac is not None
               else self._new(N, self.n, self.m)) if want_jac else None
        self._chk(cost, (N,), "cost")
        gp = self._chk(grad, (N, self.n), "grad", optional=True)
        cp = self._chk(con, (N, self.m), "con", optional=True)
        jp = self._chk(jac, (N, self.n, self.m), "con_jac", optional=True)
        _ffi.check(self._ctx, self._lib.mpc_eval_fd(self._ctx, C.byref(scc), xp, cost.data_ptr(), gp, cp, jp,
                                                    self._stream()))
        return EvalResult(cost, grad, con, jac)

    def eval_fd_static(self, sc: ScenarioBatch, x: torch.Tensor, out: EvalResult) -> EvalResult:
        """mpc_eval_fd_static (include/mpc_amd.h): eval_fd for callers that cannot allocate.  Every output
        buffer comes from the caller (``out`` must hold cost, grad, con and con_jac); the library keeps the
        per-rollout costs in its static device array, so nothing is allocated on either side of the boundary.
        At most ``_ffi.STATIC_SCRATCH_DOUBLES // (n + 1)`` scenarios; calls on one GPU must not overlap."""
        N = sc.count
        scc = self._scen_c(sc)
        xp = self._chk(x, (N, self.n), "x")
        cp0 = self._chk(out.cost, (N,), "cost")
        gp = self._chk(out.grad, (N, self.n), "grad", optional=True)
        cp = self._chk(out.con, (N, self.m), "con", optional=True)
        jp = self._chk(out.con_jac, (N, self.n, self.m), "con_jac", optional=True)
        _ffi.check(self._ctx, self._lib.mpc_eval_fd_static(self._ctx, C.byref(scc), xp, cp0, gp, cp, jp, self._stream()))
        return out

    @property
    def nz_count(self) -> int:
        return int(self._lib.mpc_jacobian_nz_count(self.H))

    def nz_supported(self) -> bool:
        """Does this context serve the structured dc/dx (it needs the prefix kernel)?"""
        empty = _ffi.MpcScenariosC()
        empty.count = 0
        return self._lib.mpc_eval_fd_nz(self._ctx, C.byref(empty), None, None, None, None, None, None) == 0

    def eval_fd_nz(self, sc: ScenarioBatch, x: torch.Tensor, want_grad: bool = True, want_con: bool = True,
                   out: Optional[EvalResult] = None) -> EvalResult:
        """Like eval_fd, with dc/dx in its structured form (include/mpc_amd.h): [N, nz_count]."""
        N = sc.count
        scc = self._scen_c(sc)
        xp = self._chk(x, (N, self.n), "x")
        cost = out.cost if out is not None else self._new(N)
        grad = (out.grad if out is not None and out.grad is not None else self._new(N, self.n)) if want_grad else None
        con = (out.con if out is not None and out.con is not None else self._new(N, self.m)) if want_con else None
        nz = out.con_jac_nz if out is not None and out.con_jac_nz is not None else self._new(N, self.nz_count)
        self._chk(cost, (N,), "cost")
        gp = self._chk(grad, (N, self.n), "grad", optional=True)
        cp = self._chk(con, (N, self.m), "con", optional=True)
        zp = self._chk(nz, (N, self.nz_count), "con_jac_nz")
        _ffi.check(self._ctx, self._lib.mpc_eval_fd_nz(self._ctx, C.byref(scc), xp, cost.data_ptr(), gp, cp, zp,
                                                       self._stream()))
        return EvalResult(cost, grad, con, None, nz)

    def expand_jacobian(self, nz: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Structured [N, nz_count] -> dense [N, n, m] on the device."""
        N = int(nz.shape[0])
        zp = self._chk(nz, (N, self.nz_count), "con_jac_nz")
        jac = out if out is not None else self._new(N, self.n, self.m)
        jp = self._chk(jac, (N, self.n, self.m), "con_jac")
        _ffi.check(self._ctx, self._lib.mpc_expand_jacobian(self._ctx, N, zp, jp, self._stream()))
        return jac

    def prepare_tick(self, q: torch.Tensor, ee_pose: torch.Tensor, base: torch.Tensor, target: torch.Tensor,
                     dq: Optional[torch.Tensor] = None, prev_solution: Optional[torch.Tensor] = None,
                     has_solution: Optional[torch.Tensor] = None, u_prev: Optional[torch.Tensor] = None,
                     has_prev: Optional[torch.Tensor] = None):
        """What ``compute_control`` assembles before ``minimize`` (ur_mpc.py:104-185), on the device:
        returns (ScenarioBatch, u_init [N,6], x0 [N,n], active [N] uint8)."""
        N = int(q.shape[0])
        qp = self._chk(q, (N, 6), "q")
        dqp = self._chk(dq, (N, 6), "dq", optional=True)
        ep = self._chk(ee_pose, (N, 6), "ee_pose")
        bp = self._chk(base, (N, 12), "base")
        self._chk(target, (N, 6), "target")
        pp = self._chk(prev_solution, (N, self.n), "prev_solution", optional=True)
        hp = self._chk(has_solution, (N,), "has_solution", dtype=torch.uint8, optional=True)
        sc = ScenarioBatch(self._new(N, 18), target, self._new(N, self.H, 12), self._new(N, 36), u_prev, has_prev)
        u_init, x0 = self._new(N, 6), self._new(N, self.n)
        active = torch.empty(N, dtype=torch.uint8, device=self.device)
        dyn = self.params.base_dyn_array()
        _ffi.check(self._ctx, self._lib.mpc_prepare_tick(
            self._ctx, N, qp, dqp, ep, bp, dyn.ctypes.data_as(C.POINTER(C.c_double)), pp, hp, sc.state0.data_ptr(),
            sc.base_traj.data_ptr(), sc.Jc.data_ptr(), u_init.data_ptr(), x0.data_ptr(), active.data_ptr(),
            self._stream()))
        return sc, u_init, x0, active

    def qp_solve(self, l: torch.Tensor, g: torch.Tensor, c: torch.Tensor, nz: torch.Tensor, lo: torch.Tensor,
                 hi: torch.Tensor):
        """SLSQP's QP subproblem for a batch (include/mpc_amd.h, mpc_qp_solve): returns (s [N,n], y [N,m+2n],
        mode [N] int32)."""
        N = int(g.shape[0])
        ldl = (self.n + 1) * self.n // 2 + 1
        lp = self._chk(l, (N, ldl), "l")
        gp = self._chk(g, (N, self.n), "g")
        cp = self._chk(c, (N, self.m), "c")
        zp = self._chk(nz, (N, self.nz_count), "nz")
        lop = self._chk(lo, (N, self.n), "lo")
        hip_ = self._chk(hi, (N, self.n), "hi")
        s = self._new(N, self.n)
        y = self._new(N, self.m + 2 * self.n)
        mode = torch.zeros(N, dtype=torch.int32, device=self.device)
        _ffi.check(self._ctx, self._lib.mpc_qp_solve(self._ctx, N, lp, gp, cp, zp, lop, hip_, s.data_ptr(), y.data_ptr(),
                                                     mode.data_ptr(), self._stream()))
        return s, y, mode

    def gather(self, src: ScenarioBatch, index: torch.Tensor, out: ScenarioBatch) -> ScenarioBatch:
        """Rows ``index`` (int32 device tensor [k], values in [0, src.count)) of ``src`` -> the first k
        rows of ``out`` in one launch; returns that k-row view of ``out``."""
        k = int(index.shape[0])
        if k > out.count:
            raise ValueError(f"{k} rows gathered into a batch of {out.count}")
        ip = self._chk(index, (k,), "index", dtype=torch.int32)
        dst = out.slice(0, k)
        if k:
            _ffi.check(self._ctx, self._lib.mpc_gather_scenarios(
                self._ctx, C.byref(self._scen_c(src, limit=False)), ip, k, C.byref(self._scen_c(dst)), self._stream()))
        return dst

    # ---- the steps either side of it -------------------------------------------------------
    def jacobian(self, q: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Batched ``get_jacobian`` (ur20_dynamics.py:74-87): q [N,6] -> J [N,6,6] (``out``: any [N,36]-sized buffer)."""
        N = int(q.shape[0])
        qp = self._chk(q, (N, 6), "q")
        J = self._new(N, 6, 6) if out is None else out
        if out is not None:
            self._chk(out.view(N, 36), (N, 36), "J")
        _ffi.check(self._ctx, self._lib.mpc_jacobian(self._ctx, N, qp, J.data_ptr(), self._stream()))
        return J

    def fk(self, q: torch.Tensor) -> torch.Tensor:
        """Batched tip pose [x,y,z,roll,pitch,yaw]: q [N,6] -> [N,6]."""
        N = int(q.shape[0])
        qp = self._chk(q, (N, 6), "q")
        pose = self._new(N, 6)
        _ffi.check(self._ctx, self._lib.mpc_fk(self._ctx, N, qp, pose.data_ptr(), self._stream()))
        return pose

    def sincos_probe(self, x: torch.Tensor):
        """The evaluation kernels' own sin/cos on x [N] (test hook, include/mpc_amd.h: mpc_sincos_probe)."""
        N = int(x.shape[0])
        xp = self._chk(x, (N,), "x")
        sn, cs = self._new(N), self._new(N)
        _ffi.check(self._ctx, self._lib.mpc_sincos_probe(self._ctx, N, xp, sn.data_ptr(), cs.data_ptr(), self._stream()))
        return sn, cs

    def ordered_sum_probe(self, terms: torch.Tensor, t0: torch.Tensor, first: torch.Tensor) -> torch.Tensor:
        """The solver kernels' ordered wavefront sum on terms [N,128] (test hook, include/mpc_amd.h: mpc_ordered_sum_probe)."""
        N = int(terms.shape[0])
        tp = self._chk(terms, (N, 128), "terms")
        zp = self._chk(t0, (N,), "t0")
        fp = self._chk(first, (N,), "first", dtype=torch.int32)
        out = self._new(N)
        _ffi.check(self._ctx, self._lib.mpc_ordered_sum_probe(self._ctx, N, tp, zp, fp, out.data_ptr(), self._stream()))
        return out

    def predict_base_dynamic(self, base: torch.Tensor, out: Optional[torch.Tensor] = None,
                             cmd: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Batched ``_predict_base_dynamic`` (ur_mpc.py:606-651): base [N,12] -> [N,H,12]; ``cmd`` [N,2] = the optional
        commanded [v_cmd, omega_cmd] of the platform state (ur_mpc.py:624-625; default: the current v and omega)."""
        N = int(base.shape[0])
        bp = self._chk(base, (N, 12), "base")
        cp = self._chk(cmd, (N, 2), "cmd", optional=True)
        if out is None:
            out = self._new(N, self.H, 12)
        else:
            self._chk(out, (N, self.H, 12), "base_traj")
        dyn = self.params.base_dyn_array()
        _ffi.check(self._ctx, self._lib.mpc_predict_base_dynamic_cmd(
            self._ctx, N, bp, cp, dyn.ctypes.data_as(C.POINTER(C.c_double)), out.data_ptr(), self._stream()))
        return out
