"""Drop-in for the reference's ``URMPC`` (ur20_mpc_controller/models/ur_mpc.py) with the inner
numeric loop on the GPU.

Same method names, argument meaning and error behaviour as the reference class, so that
``mpc_action_server.py`` (which does ``self.mpc = URMPC()``, reads ``self.mpc.dt`` and calls
``compute_control``; mpc_action_server.py:30,54,90-95) and ``scipy.optimize.minimize`` are
untouched.  What changes is where the numbers come from: every cost, constraint, Jacobian and
finite-difference gradient is produced by libmpc_amd.so (HIP, gfx950); this module only moves
small arrays and runs the same host-side glue as ``compute_control`` (early exit, DLS guess, warm
start, EMA, clip).  There is no CPU fallback: without the library or a GPU the constructor raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from scipy.optimize import minimize

from . import _ffi
from .batch import MpcContext, ScenarioBatch
from .params import JOINT_NAMES, MpcParams


class OptimizationError(Exception):
    """Raised when MPC optimization fails (ur_mpc.py:10-12)."""


def _pack_base(b: Dict) -> np.ndarray:
    return np.concatenate([np.asarray(b["position"], float), np.asarray(b["orientation"], float),
                           np.asarray(b["linear_velocity"], float)[:3], np.asarray(b["angular_velocity"], float)])


def _base_commands(base_state: Dict) -> Optional[np.ndarray]:
    """[v_cmd, omega_cmd] of a platform state that carries either key, each defaulting to the current body-x speed /
    yaw rate like ``dict.get`` in ur_mpc.py:624-625; None when the state has neither."""
    if "v_cmd" not in base_state and "omega_cmd" not in base_state:
        return None
    return np.array([float(base_state.get("v_cmd", base_state["linear_velocity"][0])),
                     float(base_state.get("omega_cmd", base_state["angular_velocity"][2]))])


def _unpack_base_traj(arr: np.ndarray) -> List[Dict]:
    return [{"position": r[0:3].copy(), "orientation": r[3:6].copy(), "linear_velocity": r[6:9].copy(),
             "angular_velocity": r[9:12].copy()} for r in arr.reshape(-1, 12)]


class KinematicBasePredictor:
    """The reference's kinematic platform extrapolation with its stateful acceleration filters
    (ur_mpc.py:527-601).  O(H) host-side glue that runs once per tick, like in the reference."""

    def __init__(self):
        self._st = {}

    def predict(self, current_base_state: Dict, horizon: int, dt: float) -> List[Dict]:
        st = self._st
        lin = np.asarray(current_base_state["linear_velocity"], float)
        ang = np.asarray(current_base_state["angular_velocity"], float)
        if "vel" in st:
            raw = (lin - st["vel"]) / dt
            acc = np.clip(0.5 * raw + (1 - 0.5) * st["acc"], -0.5, 0.5) if "acc" in st else raw * 0.5
            st["acc"] = acc.copy()
        else:
            acc = np.zeros(3)
        st["vel"] = lin.copy()
        if "avel" in st:
            raw = (ang - st["avel"]) / dt
            aacc = np.clip(0.9 * raw + (1 - 0.9) * st["aacc"], -0.3, 0.3) if "aacc" in st else raw * 0.5
            st["aacc"] = aacc.copy()
        else:
            aacc = np.zeros(3)
        st["avel"] = ang.copy()
        pos = np.asarray(current_base_state["position"], float)
        ori = np.asarray(current_base_state["orientation"], float)
        out = []
        for k in range(horizon):
            d = min(1.0, max(0.3, 1.0 - 0.2 * k))
            out.append({"position": pos + k * dt * lin + 0.5 * k * dt ** 2 * acc * d,
                        "orientation": ori + k * dt * ang + 0.5 * k * dt ** 2 * aacc * d,
                        "linear_velocity": lin + k * dt * acc * d,
                        "angular_velocity": ang + k * dt * aacc * d})
        return out


class GpuUR20Dynamics:
    """Mirror of ``UR20Dynamics`` (ur20_dynamics.py:9-163) backed by the device kernels."""

    def __init__(self, ctx: MpcContext, params: MpcParams):
        self._ctx = ctx
        self.n_joints = 6
        self.dt = params.dyn_dt                      # ur20_dynamics.py:19
        self.n_q = self.n_dq = 6
        self.n_ee = 6
        self.n_states = 18
        self.n_controls = 6
        self.joint_pos_limits = np.array([[-params.q_limit, params.q_limit]] * 6)   # ur20_dynamics.py:31-38
        self.joint_vel_limits = dict(zip(JOINT_NAMES, params.v_lim))                # ur20_dynamics.py:40-47

    def get_state_vector(self, joint_state: Dict, ee_pose: Dict) -> np.ndarray:
        return np.concatenate([joint_state["position"], joint_state["velocity"], ee_pose["position"],
                               ee_pose["orientation"]])

    def get_jacobian(self, q) -> np.ndarray:
        q = torch.as_tensor(np.asarray(q, dtype=np.float64).reshape(1, 6)).to(self._ctx.device)
        return self._ctx.jacobian(q)[0].cpu().numpy()

    def forward_kinematics(self, q) -> Dict:
        q = torch.as_tensor(np.asarray(q, dtype=np.float64).reshape(1, 6)).to(self._ctx.device)
        pose = self._ctx.fk(q)[0].cpu().numpy()
        return {"position": pose[:3].copy(), "orientation": pose[3:].copy()}

    def predict_next_state(self, current_state: np.ndarray, control: np.ndarray, dt: Optional[float] = None):
        if dt is None:
            dt = self.dt
        q = current_state[:6]
        twist = self.get_jacobian(q) @ control
        return np.concatenate([q + control * dt, control, current_state[12:15] + twist[:3] * dt,
                               current_state[15:18] + twist[3:] * dt])

    def check_limits(self, state: np.ndarray) -> bool:
        q, dq = state[:6], state[6:12]
        return bool(np.all(q >= self.joint_pos_limits[:, 0]) and np.all(q <= self.joint_pos_limits[:, 1])
                    and np.all(np.abs(dq) <= list(self.joint_vel_limits.values())))


class GpuURMPC:
    """``URMPC`` with the hot path on an MI355X.

    ``GpuURMPC()`` takes the shipped YAML values; pass ``params`` (e.g. ``MpcParams.from_yaml(path,
    horizon=20)``) to mirror what the ROS parameter server would hold.
    """

    def __init__(self, params: Optional[MpcParams] = None, device=0, verbose: bool = False, solver: str = "scipy"):
        """``solver="scipy"`` (default) is the reference's own call, ``scipy.optimize.minimize(method="SLSQP")`` with
        the device behind its callbacks (ur_mpc.py:188-200).  ``solver="resident"`` hands the same start point to the
        library's device-resident SLSQP (``device_solver.ResidentSLSQP``: scipy's algorithm restated, iterates
        bit-identical to scipy's compiled core) and skips the per-callback round trips: about 11 ms instead of 41 ms
        per H=20 tick, same controls to the last bit in the tests; nothing but ``minimize`` itself is replaced."""
        if solver not in ("scipy", "resident"):
            raise ValueError("solver must be 'scipy' or 'resident'")
        self.params = params if params is not None else MpcParams.reference_yaml_defaults()
        p = self.params
        self._ctx = MpcContext(p, max_scenarios=1, device=device)
        self._resident = None
        if solver == "resident":
            from .device_solver import ResidentSLSQP

            self._resident = ResidentSLSQP(self._ctx, capacity=1)
        self.dynamics = GpuUR20Dynamics(self._ctx, p)
        self.horizon, self.dt = p.horizon, p.dt
        self.w_position, self.w_orientation = p.w_position, p.w_orientation
        self.w_control, self.w_smooth, self.w_damping = p.w_control, p.w_smooth, p.w_damping
        self.w_linear, self.w_angular = p.w_linear, p.w_angular
        self.use_dynamic_model = p.use_dynamic_model
        self.previous_solution = None
        # NB: like the reference, ``previous_control`` does not exist until the first successful solve
        self._disp = verbose
        self._cache_key = None
        self._cache = None
        self.last_result = None
        self._kin = KinematicBasePredictor()
        self._make_staging()

    # ---- one tick's buffers: allocated once, so that a tick moves one small pinned block each way ------------
    def _make_staging(self):
        """Device-resident scenario of THIS controller plus pinned staging: per tick one H2D of [q, platform
        state], one D2H of Jc, one H2D of [state0, target, previous_control]; per scipy callback one H2D of x and
        one D2H of [f, grad f, c, structured dc/dx] (the dense dc/dx is filled in on the host, as the batched
        drivers do)."""
        ctx, p = self._ctx, self.params
        dev, n, m, H = ctx.device, p.n, p.m, p.horizon
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
        pin = lambda count: torch.zeros(count, dtype=torch.float64).pin_memory()
        self._qb_h, self._qb_d = pin(20), z(20)                       # [q(6), platform state(12), v_cmd, omega_cmd]
        self._jc_h = pin(36)
        self._in_h, self._in_d = pin(30), z(30)                       # [state0(18), target(6), previous_control(6)]
        self._x_h, self._x_d = pin(n), z(1, n)
        self._flags = (torch.zeros(1, dtype=torch.uint8, device=dev), torch.ones(1, dtype=torch.uint8, device=dev))
        self._bt_d, self._jc_d = z(1, H, 12), z(1, 36)
        self._structured = ctx.nz_supported()
        nj = ctx.nz_count if self._structured else n * m
        self._out_h, self._out_d = pin(1 + n + m + nj), z(1 + n + m + nj)
        o = self._out_d
        self._views = (o[0:1], o[1:1 + n].view(1, n), o[1 + n:1 + n + m].view(1, m),
                       o[1 + n + m:].view(1, nj) if self._structured else o[1 + n + m:].view(1, n, m))
        if self._structured:
            from .slsqp_state import nz_scatter_index

            self._scatter = nz_scatter_index(H)
        self._tick_sc = None

    def _tick_scenario(self, current_state, target_ee_pose) -> ScenarioBatch:
        """The resident scenario of this tick (base_traj and Jc were written by the device a moment ago)."""
        has_prev = hasattr(self, "previous_control")
        h = self._in_h.numpy()
        h[0:18] = current_state
        h[18:21], h[21:24] = target_ee_pose["position"], target_ee_pose["orientation"]
        h[24:30] = self.previous_control if has_prev else 0.0
        self._in_d.copy_(self._in_h, non_blocking=True)
        d = self._in_d
        self._tick_sc = ScenarioBatch(d[0:18].view(1, 18), d[18:24].view(1, 6), self._bt_d, self._jc_d, d[24:30].view(1, 6),
                                      self._flags[1 if has_prev else 0])
        self._tick_scc = self._ctx._scen_c(self._tick_sc)      # checked once per tick; the callbacks reuse the struct
        return self._tick_sc

    def _evaluate_tick(self, x: np.ndarray):
        """_evaluate on the resident scenario: one small copy each way per distinct x."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        key = ("tick", x.tobytes())
        if key != self._cache_key:
            if not np.all(np.isfinite(x)):
                raise ValueError("non-finite control sequence")
            ctx, p = self._ctx, self.params
            n, m = p.n, p.m
            self._x_h.numpy()[:] = x
            self._x_d.view(-1).copy_(self._x_h, non_blocking=True)
            cost, grad, con, jac = self._views
            stream = torch.cuda.current_stream(ctx.device)
            fn = ctx._lib.mpc_eval_fd_nz if self._structured else ctx._lib.mpc_eval_fd
            _ffi.check(ctx._ctx, fn(ctx._ctx, self._tick_scc, self._x_d.data_ptr(), cost.data_ptr(), grad.data_ptr(),
                                    con.data_ptr(), jac.data_ptr(), stream.cuda_stream))
            self._out_h.copy_(self._out_d, non_blocking=True)
            stream.synchronize()
            o = self._out_h.numpy()
            f = float(o[0])
            if not np.isfinite(f):
                raise ValueError("non-finite cost")      # a replacement must raise, not return NaN
            if self._structured:
                A = np.zeros((n + 1, m))
                A.reshape(-1)[self._scatter] = o[1 + n + m:]
                J = A[:n].T
            else:
                J = o[1 + n + m:].reshape(n, m).T.copy()
            self._cache = (f, o[1:1 + n].copy(), o[1 + n:1 + n + m].copy(), J)
            self._cache_key = key
        return self._cache

    # ---- device plumbing ---------------------------------------------------------------------
    def _scenario(self, current_state, target_ee_pose, base_trajectory) -> ScenarioBatch:
        bt = np.stack([_pack_base(b) for b in base_trajectory])
        if bt.shape[0] != self.horizon:
            raise ValueError("base_trajectory must have one entry per horizon step")
        has_prev = hasattr(self, "previous_control")
        target = np.concatenate([target_ee_pose["position"], target_ee_pose["orientation"]])
        return ScenarioBatch.from_numpy(
            self._ctx.device, np.asarray(current_state, float), target, bt, np.asarray(self._J_cached, float),
            u_prev=self.previous_control if has_prev else np.zeros(6), has_prev=np.array([1 if has_prev else 0]))

    def _evaluate(self, x: np.ndarray, sc: ScenarioBatch):
        """f, c, FD grad, FD dc/dx at x; one launch serves the four callbacks scipy makes at one x."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        key = (id(sc), x.tobytes())
        if key != self._cache_key:
            if not np.all(np.isfinite(x)):
                raise ValueError("non-finite control sequence")
            xd = torch.as_tensor(x.reshape(1, -1)).to(self._ctx.device)
            r = self._ctx.eval_fd(sc, xd)
            f = float(r.cost.cpu()[0])
            if not np.isfinite(f):
                raise ValueError("non-finite cost")      # a replacement must raise, not return NaN
            self._cache = (f, r.grad[0].cpu().numpy(), r.con[0].cpu().numpy(), r.con_jac[0].cpu().numpy().T)
            self._cache_key = key
        return self._cache

    # ---- the reference's callables ---------------------------------------------------------------
    def _cost_function(self, x: np.ndarray, current_state: np.ndarray, target_ee_pose: Dict,
                       base_trajectory: List[Dict]) -> float:
        """ur_mpc.py:665-692 (needs ``_J_cached`` like the reference's stage cost, ur_mpc.py:252)."""
        sc = self._scenario(current_state, target_ee_pose, base_trajectory)
        xd = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64).reshape(1, -1)).to(self._ctx.device)
        return float(self._ctx.eval(sc, xd, want_con=False).cost.cpu()[0])

    def _get_constraints(self, current_state: np.ndarray, base_trajectory: List[Dict]) -> List[Dict]:
        """ur_mpc.py:383-418: one 'ineq' constraint, c(x) in R^{12(H+1)}.  The constraint reads only
        the joint states, so a zero target and the given platform trajectory are enough."""
        if not hasattr(self, "_J_cached"):
            self._J_cached = self.dynamics.get_jacobian(current_state[:6])
        sc = self._scenario(current_state, {"position": np.zeros(3), "orientation": np.zeros(3)}, base_trajectory)
        return [{"type": "ineq", "fun": lambda x: self._evaluate(x, sc)[2].copy(),
                 "jac": lambda x: self._evaluate(x, sc)[3]}]

    def _get_bounds(self, x0: np.ndarray) -> List[Tuple[float, float]]:
        """ur_mpc.py:420-430"""
        return self.params.bounds()

    def _world_to_base(self, vec_w: np.ndarray, yaw: float) -> np.ndarray:
        """ur_mpc.py:714-720"""
        c, s = np.cos(yaw), np.sin(yaw)
        return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]]) @ vec_w

    def predict_trajectory(self, initial_state: np.ndarray, control_sequence: np.ndarray, horizon: int):
        """ur_mpc.py:694-710: H+1 states; the H Jacobians come from one batched device call."""
        U = np.asarray(control_sequence, float).reshape(horizon, 6)
        qs = [initial_state[:6]]
        for k in range(horizon):       # sequential adds, like predict_next_state
            qs.append(qs[-1] + U[k] * self.dt)
        q = np.stack(qs)
        J = self._ctx.jacobian(torch.as_tensor(q[:horizon]).to(self._ctx.device)).cpu().numpy()
        out = [initial_state]
        s = initial_state.copy()
        for k in range(horizon):
            tw = J[k] @ U[k]
            s = np.concatenate([q[k + 1], U[k], s[12:15] + tw[:3] * self.dt, s[15:18] + tw[3:] * self.dt])
            out.append(s)
        return out

    # ---- platform predictors ---------------------------------------------------------------------
    def _predict_base_dynamic(self, current_base_state: Dict, horizon: int) -> List[Dict]:
        """ur_mpc.py:606-651 on the device, the optional commanded velocities 'v_cmd' / 'omega_cmd' of the
        actuator-lag model included (ur_mpc.py:624-625; the odometry dict the server passes has neither,
        base_observer.py:55-62)."""
        if horizon != self.horizon:
            raise ValueError("horizon is fixed per controller")
        b = torch.as_tensor(_pack_base(current_base_state).reshape(1, 12)).to(self._ctx.device)
        cmd = _base_commands(current_base_state)
        cmd_d = None if cmd is None else torch.as_tensor(cmd.reshape(1, 2)).to(self._ctx.device)
        return _unpack_base_traj(self._ctx.predict_base_dynamic(b, cmd=cmd_d)[0].cpu().numpy())

    def _predict_base_kinematic(self, current_base_state: Dict, horizon: int) -> List[Dict]:
        """ur_mpc.py:527-601 (host glue, stateful)."""
        return self._kin.predict(current_base_state, horizon, self.dt)

    def predict_base_motion(self, current_base_state: Dict, horizon: int) -> List[Dict]:
        """ur_mpc.py:655-663"""
        if self.use_dynamic_model:
            return self._predict_base_dynamic(current_base_state, horizon)
        return self._predict_base_kinematic(current_base_state, horizon)

    def _minimize_resident(self, x0: np.ndarray):
        """What ``minimize`` returns that compute_control reads (x, success, message, fun, nit, status), from the
        device-resident solver on this tick's resident scenario."""
        from scipy.optimize import OptimizeResult

        from .slsqp_state import EXIT_MESSAGES

        r = self._resident.solve(self._tick_sc, np.asarray(x0, dtype=np.float64).reshape(1, -1))
        status = int(r.status[0])
        return OptimizeResult(x=r.x[0].copy(), fun=float(r.fun[0]), nit=int(r.nit[0]), status=status, success=status == 0,
                              message=EXIT_MESSAGES.get(status, f"exit mode {status}"))

    # ---- one control tick ------------------------------------------------------------------------
    def compute_control(self, current_joint_state: Dict, current_ee_pose: Dict, target_ee_pose: Dict,
                        base_state: Dict) -> np.ndarray:
        """ur_mpc.py:83-247, same control flow and fallbacks."""
        initial_vel = None
        try:
            if (np.linalg.norm(base_state["linear_velocity"]) < 0.01
                    and np.linalg.norm(base_state["angular_velocity"]) < 0.01):
                return np.zeros(6)                                            # ur_mpc.py:108-109
            lever = current_ee_pose["position"] - base_state["position"]
            v_ind = base_state["linear_velocity"] + np.cross(base_state["angular_velocity"], lever)
            yaw = base_state["orientation"][2]
            tl = self._world_to_base(-v_ind, yaw)
            ta = self._world_to_base(-base_state["angular_velocity"], yaw)
            current_state = self.dynamics.get_state_vector(current_joint_state, current_ee_pose)
            # platform prediction and Jc = J(q) are made on the device straight into this controller's resident
            # scenario (ur_mpc.py:154,157); only Jc comes back, for the DLS start
            ctx = self._ctx
            qb = self._qb_h.numpy()
            qb[0:6] = current_joint_state["position"]
            cmd = None
            if self.use_dynamic_model:
                qb[6:18] = _pack_base(base_state)
                cmd = _base_commands(base_state)                              # ur_mpc.py:624-625
                if cmd is not None:
                    qb[18:20] = cmd
            self._qb_d.copy_(self._qb_h, non_blocking=True)
            ctx.jacobian(self._qb_d[0:6].view(1, 6), out=self._jc_d)
            if self.use_dynamic_model:
                ctx.predict_base_dynamic(self._qb_d[6:18].view(1, 12), out=self._bt_d,
                                         cmd=None if cmd is None else self._qb_d[18:20].view(1, 2))
            else:
                bt = np.stack([_pack_base(b) for b in self._predict_base_kinematic(base_state, self.horizon)])
                self._bt_d.copy_(torch.from_numpy(bt).view(1, self.horizon, 12))
            self._jc_h.copy_(self._jc_d.view(-1), non_blocking=True)
            torch.cuda.current_stream(ctx.device).synchronize()
            self._J_cached = self._jc_h.numpy().reshape(6, 6).copy()
            J = np.vstack([self._J_cached[:3, :], self._J_cached[3:, :]])
            pinv = J.T @ np.linalg.inv(J @ J.T + 0.01 * np.eye(6))            # ur_mpc.py:173-175
            initial_vel = pinv @ np.concatenate([tl, ta])
            if self.previous_solution is not None:                            # ur_mpc.py:178-185
                x0 = np.roll(self.previous_solution, -6)
                x0[-6:] = initial_vel
            else:
                x0 = np.tile(initial_vel, self.horizon)
            self._tick_scenario(current_state, target_ee_pose)
            self._cache_key = None
            ev = self._evaluate_tick
            result = self._minimize_resident(x0) if self._resident is not None else minimize(
                fun=lambda x: ev(x)[0], x0=x0, method="SLSQP",
                jac=lambda x: ev(x)[1],                                       # scipy's own FD, made on the device
                bounds=self._get_bounds(x0),
                constraints=[{"type": "ineq", "fun": lambda x: ev(x)[2].copy(), "jac": lambda x: ev(x)[3]}],
                options={"ftol": self.params.ftol, "maxiter": self.params.maxiter, "disp": self._disp})
            self.last_result = result
            if not result.success:
                raise OptimizationError(f"Optimization failed: {result.message}")
            control = result.x[:6]
            if hasattr(self, "previous_control"):
                control = 0.5 * control + (1.0 - 0.5) * self.previous_control   # ur_mpc.py:209-212
            self.previous_control = control.copy()
            self.previous_solution = result.x
            return np.clip(control, -0.5, 0.5)                                # ur_mpc.py:234-235
        except OptimizationError:
            return initial_vel
        except ValueError:
            return np.zeros(6)
        except _ffi.MpcError:
            raise                      # a failing device is not an "unexpected input": never mask it
        except Exception:
            # like the reference (ur_mpc.py:245-247) an unexpected failure yields a zero command
            return np.zeros(6)
