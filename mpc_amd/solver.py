"""Batched SLSQP: N independent MPC problems advanced in lock-step.

The reference solves one problem per tick with ``scipy.optimize.minimize(method='SLSQP')``
(ur_mpc.py:188-200).  scipy's driver (``_slsqp_py.py:216-456``) is a reverse-communication loop
around a Fortran core: the core returns ``mode = 1`` ("give me f and c at x") or ``mode = -1``
("give me grad f and dc/dx at x") until it is done.  Here every problem keeps its own core state
on the host, and each round's requests are served by ONE device launch per request kind
(``mpc_eval`` for the mode-1 problems, ``mpc_eval_fd`` for the mode -1 problems).  The iterates are
those of scipy's own driver with the reference's options (``ftol=1e-3, maxiter=50, eps=1e-3``),
because the same core sees the same numbers.

The per-problem quadratic subproblem stays on the host (scipy's compiled core, n = 6H variables,
m = 12(H+1) inequality rows): solves/s is host-bound, evals/s is not.  The dense dc/dx arrives in
exactly the memory layout the core wants (column-major m x (n+1) with a zero last column,
``_slsqp_py.py:509``), so it is one D2H copy into the core's own array.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .batch import EvalResult, MpcContext, ScenarioBatch

from .slsqp_state import EXIT_MESSAGES, Problem, nz_scatter_index, require_core  # noqa: F401


@dataclass
class BatchSolveResult:
    x: np.ndarray          # [N, n]
    fun: np.ndarray        # [N]
    nit: np.ndarray        # [N]
    status: np.ndarray     # [N] SLSQP exit mode, 0 = success
    rounds: int            # lock-step rounds = device launches of each kind at most
    n_eval: int            # scenario-evals served by mpc_eval   (f, c)
    n_eval_fd: int         # scenario-evals served by mpc_eval_fd (f, c, grad, dc/dx)
    t_core: float = 0.0    # seconds inside the SLSQP core (host)
    t_serve: float = 0.0   # seconds in device launches + transfers

    @property
    def success(self):
        return self.status == 0


class BatchedSLSQP:
    def __init__(self, ctx: MpcContext):
        require_core()
        self.ctx = ctx
        p = ctx.params
        self.n, self.m = p.n, p.m
        self.lb = np.array([b[0] for b in p.bounds()])
        self.ub = np.array([b[1] for b in p.bounds()])

    def _to_dev(self, a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.ctx.device)

    def solve(self, sc: ScenarioBatch, x0: np.ndarray) -> BatchSolveResult:
        ctx, n, m = self.ctx, self.n, self.m
        N = sc.count
        if N > ctx.max_scenarios:
            raise ValueError("batch larger than the context")
        X = np.clip(np.array(x0, dtype=np.float64).reshape(N, n), self.lb, self.ub)   # _slsqp_py.py:268
        A = np.zeros((N, n + 1, m))          # A[i].T is the core's a(la, n+1), column-major
        structured = ctx.nz_supported()      # dc/dx crosses PCIe in its structured form where the kernel offers it
        Aflat = A.reshape(N, -1)
        scatter = nz_scatter_index(ctx.params.horizon) if structured else None
        probs: List[Problem] = [Problem(X[i], n, m, ctx.params.ftol, ctx.params.maxiter, A[i].T)
                                 for i in range(N)]
        xl, xu = self.lb.copy(), self.ub.copy()
        sub_buf = ScenarioBatch.empty(ctx.device, N, ctx.params.horizon, sc.u_prev is not None)
        n_eval = n_eval_fd = rounds = 0
        t_core = t_serve = 0.0

        prof = self.profile = {"gather": 0.0, "upload": 0.0, "kernel": 0.0, "d2h": 0.0, "scatter": 0.0}
        profiling = bool(os.environ.get("MPC_AMD_SOLVE_PROFILE"))

        def tick(key, t):
            if profiling:
                torch.cuda.synchronize(ctx.device)
                now = time.perf_counter()
                prof[key] += now - t
                return now
            return t

        def serve(idx: np.ndarray, need_grad: bool):
            """One device launch for the problems in idx; x is clipped like _clip_x_for_func."""
            t = time.perf_counter()
            if len(idx) == N:
                sub = sc
            else:
                sub = ctx.gather(sc, torch.from_numpy(idx.astype(np.int32)).to(ctx.device), sub_buf)
            t = tick("gather", t)
            xd = self._to_dev(np.clip(X[idx], self.lb, self.ub))
            t = tick("upload", t)
            if need_grad and structured:
                r = ctx.eval_fd_nz(sub, xd)
                t = tick("kernel", t)
                g = r.grad.cpu().numpy()
                nz = r.con_jac_nz.cpu().numpy()
                t = tick("d2h", t)
                for k, i in enumerate(idx):
                    probs[i].g[:n] = g[k]
                    probs[i].g[n] = 0.0
                Aflat[np.asarray(idx)[:, None], scatter[None, :]] = nz     # every other entry stays zero
                tick("scatter", t)
                return r
            if need_grad:
                r = ctx.eval_fd(sub, xd)
                t = tick("kernel", t)
                g = r.grad.cpu().numpy()
                jac = r.con_jac.cpu().numpy()
                t = tick("d2h", t)
                for k, i in enumerate(idx):
                    probs[i].g[:n] = g[k]
                    probs[i].g[n] = 0.0
                    A[i, :n, :] = jac[k]
                tick("scatter", t)
                return r
            r = ctx.eval(sub, xd)
            tick("kernel", t)
            return r

        # mode 0 on entry: f, g, c, a at the start point (_slsqp_py.py:419-425)
        allidx = np.arange(N)
        t0 = time.perf_counter()
        r = serve(allidx, True)
        f0 = r.cost.cpu().numpy()
        c0 = r.con.cpu().numpy()
        for i in allidx:
            probs[i].fx = float(f0[i])
            probs[i].c[:] = c0[i]
        n_eval_fd += N
        t_serve += time.perf_counter() - t0
        active = list(range(N))
        while active:
            rounds += 1
            want_f, want_g = [], []
            t0 = time.perf_counter()
            for i in active:
                md = probs[i].step(xl, xu)
                if md == 1:
                    want_f.append(i)
                elif md == -1:
                    want_g.append(i)
            t1 = time.perf_counter()
            t_core += t1 - t0
            if want_f:
                idx = np.array(want_f)
                r = serve(idx, False)
                f = r.cost.cpu().numpy()
                c = r.con.cpu().numpy()
                for k, i in enumerate(idx):
                    probs[i].fx = float(f[k])
                    probs[i].c[:] = c[k]
                n_eval += len(idx)
            if want_g:
                serve(np.array(want_g), True)
                n_eval_fd += len(want_g)
            t_serve += time.perf_counter() - t1
            active = [i for i in active if not probs[i].done]
        return BatchSolveResult(
            x=X, fun=np.array([p.fx for p in probs]), nit=np.array([int(p.majiter) for p in probs]),
            status=np.array([int(p.mode) for p in probs]), rounds=rounds, n_eval=n_eval, n_eval_fd=n_eval_fd,
            t_core=t_core, t_serve=t_serve)


def _gather(sc: ScenarioBatch, idx: np.ndarray, device) -> ScenarioBatch:
    """Rows idx of a device-resident batch (contiguous copy; a full ascending range is returned as is)."""
    if len(idx) == sc.count and np.array_equal(idx, np.arange(sc.count)):
        return sc
    t = torch.as_tensor(idx, dtype=torch.long, device=device)
    f = lambda a: None if a is None else a.index_select(0, t).contiguous()
    return ScenarioBatch(f(sc.state0), f(sc.target), f(sc.base_traj), f(sc.Jc), f(sc.u_prev), f(sc.has_prev))


class BatchedKinematicPredictor:
    """``_predict_base_kinematic`` (ur_mpc.py:527-601) for N controllers: the same elementwise
    arithmetic as ``mpc_amd.ur_mpc.KinematicBasePredictor`` row by row, with each controller's
    acceleration filters kept as rows (NaN = "no sample yet").  Only the controllers that predict on
    a tick (``rows``) update their filters, as in the reference, where the early exit comes first."""

    def __init__(self, count: int):
        self.vel = np.full((count, 3), np.nan)
        self.acc = np.full((count, 3), np.nan)
        self.avel = np.full((count, 3), np.nan)
        self.aacc = np.full((count, 3), np.nan)

    @staticmethod
    def _filter(cur, prev, prev_acc, alpha, clip, dt):
        acc = np.zeros_like(cur)
        seen = ~np.isnan(prev[:, 0])
        have_acc = seen & ~np.isnan(prev_acc[:, 0])
        raw = (cur - np.where(seen[:, None], prev, 0.0)) / dt
        first = seen & ~have_acc
        acc[first] = raw[first] * 0.5
        acc[have_acc] = np.clip(alpha * raw[have_acc] + (1 - alpha) * prev_acc[have_acc], -clip, clip)
        new_acc = prev_acc.copy()
        new_acc[seen] = acc[seen]
        return acc, new_acc

    def predict(self, rows: np.ndarray, base: np.ndarray, horizon: int, dt: float) -> np.ndarray:
        """base [k,12] of controllers `rows` -> [k,H,12]."""
        lin, ang = base[:, 6:9], base[:, 9:12]
        acc, self.acc[rows] = self._filter(lin, self.vel[rows], self.acc[rows], 0.5, 0.5, dt)
        self.vel[rows] = lin
        aacc, self.aacc[rows] = self._filter(ang, self.avel[rows], self.aacc[rows], 0.9, 0.3, dt)
        self.avel[rows] = ang
        out = np.empty((base.shape[0], horizon, 12))
        for k in range(horizon):
            d = min(1.0, max(0.3, 1.0 - 0.2 * k))
            out[:, k, 0:3] = base[:, 0:3] + k * dt * lin + 0.5 * k * dt ** 2 * acc * d
            out[:, k, 3:6] = base[:, 3:6] + k * dt * ang + 0.5 * k * dt ** 2 * aacc * d
            out[:, k, 6:9] = lin + k * dt * acc * d
            out[:, k, 9:12] = ang + k * dt * aacc * d
        return out


class BatchedMPC:
    """N controllers ticking together: the batched equivalent of ``URMPC.compute_control``
    (ur_mpc.py:83-247) including the early exit, the DLS start, warm start, output EMA and clip."""

    def __init__(self, ctx: MpcContext, solver=None, prep: str = "host"):
        """`solver`: anything with ``solve(ScenarioBatch, x0) -> BatchSolveResult`` on this context
        (default: the in-process BatchedSLSQP; pass a ``mpc_amd.pool.SolverPool`` to use more cores).
        `prep`: where the per-tick inputs are assembled.  "host" forms the DLS start with numpy in
        the reference's operation order, J^T inv(J J^T + 0.01 I) twist (start points bit-compatible with it); "device" does everything in one
        launch of ``mpc_prepare_tick`` (start points equal to rounding; for very large batches); "resident" is
        "device" with nothing but the tick's inputs and outputs crossing PCIe: the start points go from
        ``mpc_prepare_tick`` straight into the device-resident solve (needs a solver with ``device_io``:
        ``ResidentSLSQP``), the solutions stay on the device as the next tick's warm start, and only the first control
        of each comes back (``previous_solution`` fetches them when asked)."""
        if prep not in ("host", "device", "resident"):
            raise ValueError("prep must be 'host', 'device' or 'resident'")
        if prep == "resident" and not getattr(solver, "device_io", False):
            raise ValueError("prep='resident' needs a solver that keeps start points and solutions on the device (ResidentSLSQP)")
        self.ctx = ctx
        self.prep = prep
        self.solver = solver if solver is not None else BatchedSLSQP(ctx)
        self._prev_sol: Optional[np.ndarray] = None           # [N, n], NaN rows = no solution yet
        self._prev_sol_dev: Optional[torch.Tensor] = None     # prep="resident": the same on the device (rows without: zeros)
        self._have_sol: Optional[np.ndarray] = None           # prep="resident": which rows hold a solution
        self.previous_control: Optional[np.ndarray] = None    # [N, 6], NaN rows = attribute absent
        self.last: Optional[BatchSolveResult] = None     # the solve of the latest tick; None if that tick solved nothing
        self.last_rows: np.ndarray = np.zeros(0, dtype=np.int64)   # controllers (rows) `last` refers to
        self._kin: Optional[BatchedKinematicPredictor] = None

    @property
    def previous_solution(self) -> Optional[np.ndarray]:
        """[N, n] the last accepted solution of every controller, NaN rows where there is none yet (ur_mpc.py:178-185)."""
        if self.prep == "resident" and self._prev_sol_dev is not None:
            out = self._prev_sol_dev.cpu().numpy()
            out[~self._have_sol] = np.nan
            return out
        return self._prev_sol

    @previous_solution.setter
    def previous_solution(self, value):
        self._prev_sol = value

    def compute_control(self, q: np.ndarray, ee_pose: np.ndarray, target: np.ndarray, base: np.ndarray) -> np.ndarray:
        """q [N,6], ee_pose [N,6] = [pos, rpy], target [N,6], base [N,12] -> controls [N,6]."""
        from .scenarios import dls_guess

        ctx, p = self.ctx, self.ctx.params
        N, n, H = q.shape[0], p.n, p.horizon
        if self.previous_control is None:
            self._prev_sol = np.full((N, n), np.nan) if self.prep != "resident" else None
            self.previous_control = np.full((N, 6), np.nan)
            if self.prep == "resident":
                self._prev_sol_dev = torch.zeros(N, n, dtype=torch.float64, device=ctx.device)
                self._have_sol = np.zeros(N, dtype=bool)
        out = np.zeros((N, 6))
        active = ~((np.linalg.norm(base[:, 6:9], axis=1) < 0.01) & (np.linalg.norm(base[:, 9:12], axis=1) < 0.01))
        idx = np.nonzero(active)[0]
        self.last, self.last_rows = None, idx        # `last` belongs to this tick (None: every controller exited early)
        if len(idx) == 0:
            return out
        dev = ctx.device
        up = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev)
        bt_kin = None
        if not p.use_dynamic_model:                                        # ur_mpc.py:660-663
            if self._kin is None:
                self._kin = BatchedKinematicPredictor(N)
            bt_kin = up(self._kin.predict(idx, base[idx], H, p.dt))
        everyone = len(idx) == N
        if self.prep == "resident":
            prev_sol, have_sol = None, self._have_sol[idx]
        else:
            prev_sol = self._prev_sol[idx]
            have_sol = ~np.isnan(prev_sol[:, 0])
        prev_u = self.previous_control[idx]
        have_u = ~np.isnan(prev_u[:, 0])
        u_prev_d = up(np.nan_to_num(prev_u))
        have_u_d = torch.as_tensor(have_u.astype(np.uint8)).to(dev)
        if self.prep == "resident":
            idx_d = None if everyone else torch.as_tensor(idx).to(dev)
            sc, u_init_d, x0_d, _ = ctx.prepare_tick(
                up(q[idx]), up(ee_pose[idx]), up(base[idx]), up(target[idx]),
                prev_solution=self._prev_sol_dev if everyone else self._prev_sol_dev[idx_d].contiguous(),
                has_solution=torch.as_tensor(have_sol.astype(np.uint8)).to(dev), u_prev=u_prev_d, has_prev=have_u_d)
            if bt_kin is not None:
                sc.base_traj = bt_kin
            res = self.solver.solve(sc, x0_d, x_on_device=True)
            self.last = res
            ok = res.success
            u = res.x_dev[:, :6].cpu().numpy()                                  # 48 of the 8 n bytes of a solution
            u_init = u_init_d.cpu().numpy()
            u[have_u & ok] = 0.5 * u[have_u & ok] + 0.5 * prev_u[have_u & ok]      # ur_mpc.py:209-212
            ctrl = np.where(ok[:, None], np.clip(u, -0.5, 0.5), u_init)            # ur_mpc.py:234-241
            self.previous_control[idx[ok]] = u[ok]
            if ok.all() and everyone:
                self._prev_sol_dev.copy_(res.x_dev)
            elif ok.any():
                rows_d = torch.as_tensor(idx[ok]).to(dev)
                self._prev_sol_dev[rows_d] = res.x_dev[torch.as_tensor(np.nonzero(ok)[0]).to(dev)]
            self._have_sol[idx[ok]] = True
            out[idx] = ctrl
            return out
        if self.prep == "device":
            sc, u_init_d, x0_d, _ = ctx.prepare_tick(
                up(q[idx]), up(ee_pose[idx]), up(base[idx]), up(target[idx]), prev_solution=up(np.nan_to_num(prev_sol)),
                has_solution=torch.as_tensor(have_sol.astype(np.uint8)).to(dev), u_prev=u_prev_d, has_prev=have_u_d)
            u_init, x0 = u_init_d.cpu().numpy(), x0_d.cpu().numpy()
            if bt_kin is not None:
                sc.base_traj = bt_kin
        else:
            Jc = ctx.jacobian(up(q[idx])).cpu().numpy().reshape(-1, 36)
            bt = bt_kin if bt_kin is not None else ctx.predict_base_dynamic(up(base[idx]))
            u_init = dls_guess(Jc, base[idx], ee_pose[idx, :3])
            x0 = np.tile(u_init, (1, H))
            if have_sol.any():                                             # ur_mpc.py:178-181
                rolled = np.roll(prev_sol[have_sol], -6, axis=1)
                rolled[:, -6:] = u_init[have_sol]
                x0[have_sol] = rolled
            state0 = np.concatenate([q[idx], np.zeros((len(idx), 6)), ee_pose[idx]], axis=1)
            sc = ScenarioBatch(up(state0), up(target[idx]), bt, up(Jc), u_prev_d, have_u_d)
        res = self.solver.solve(sc, x0)
        self.last = res
        ok = res.success
        u = res.x[:, :6].copy()
        u[have_u & ok] = 0.5 * u[have_u & ok] + 0.5 * prev_u[have_u & ok]      # ur_mpc.py:209-212
        ctrl = np.where(ok[:, None], np.clip(u, -0.5, 0.5), u_init)            # ur_mpc.py:234-241
        self.previous_control[idx[ok]] = u[ok]
        self._prev_sol[idx[ok]] = res.x[ok]
        out[idx] = ctrl
        return out
