"""TEST INFRASTRUCTURE — numpy restatement of the reference's MPC inner loop.

This file is the *oracle*, not the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``mpc_amd/`` does.

It restates, in reference-shaped numpy (same numpy calls on the same small
arrays, a Python loop over the horizon, scipy's own ``approx_derivative`` for
the finite differences), the path that ``URMPC.compute_control`` hands to
``scipy.optimize.minimize``.  File:line citations are relative to
``/root/reference/ur20_mpc_controller/ur20_mpc_controller/models/``:

* ``ur_mpc.py``        -> MPC:<line>
* ``ur20_dynamics.py`` -> DYN:<line>

Pinning: ``tests/test_oracle_vs_reference.py`` imports the real reference
(under ROS stubs, see ``oracle/ref_shim.py``) when ``/root/reference`` exists
and compares bit-for-bit; ``tests/golden/*.npz`` hold vectors generated from
the real reference by ``oracle/gen_golden.py`` for boxes where it is absent.

The one thing the reference does NOT contain is the Jacobian arithmetic: it
calls MoveIt (DYN:82).  ``ur20_jacobian`` below is this repo's own UR20
geometric Jacobian ("parity unpinned" at that boundary); the
golden vectors were generated with this same provider injected into the
reference through the stubbed ``MoveGroupCommander.get_jacobian_matrix``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

# --------------------------------------------------------------------------
# UR20 kinematic data table (nominal DH, general knowledge; unverifiable here)
# --------------------------------------------------------------------------


@dataclass
class KinTable:
    """Standard-DH table of a 6R arm plus the fixed transforms either side.

    ``ca``/``sa`` hold cos/sin of the link twists as exact 0/+-1 values so that
    no 6e-17 residue of cos(pi/2) leaks into the Jacobian.  ``base_c/base_s``
    is the yaw of the DH base relative to the frame the Jacobian is expressed
    in (``base_link`` in ur_description is rotated pi about z relative to
    ``base_link_inertia``).  ``tool`` is the translation from the DH end frame
    (``tool0``) to the tip link, expressed in the end frame.
    """

    a: Sequence[float] = (0.0, -0.8620, -0.7287, 0.0, 0.0, 0.0)
    d: Sequence[float] = (0.2363, 0.0, 0.0, 0.2010, 0.1593, 0.1543)
    ca: Sequence[float] = (0.0, 1.0, 1.0, 0.0, 0.0, 1.0)
    sa: Sequence[float] = (1.0, 0.0, 0.0, 1.0, -1.0, 0.0)
    base_c: float = -1.0
    base_s: float = 0.0
    tool: Sequence[float] = (0.0, 0.0, 0.02)


UR20_KIN = KinTable()


def _chain(q, kin: KinTable):
    """Walk the DH chain; returns joint axes, joint origins, tip rotation, tip position.

    Scalar arithmetic in a fixed order (no FMA in CPython) — the C oracle and
    the strict HIP kernel use the same order.
    """
    cb, sb = kin.base_c, kin.base_s
    R = [[cb, -sb, 0.0], [sb, cb, 0.0], [0.0, 0.0, 1.0]]
    o = [0.0, 0.0, 0.0]
    zs, os_ = [], []
    for i in range(6):
        zs.append((R[0][2], R[1][2], R[2][2]))
        os_.append((o[0], o[1], o[2]))
        c = math.cos(q[i])
        s = math.sin(q[i])
        ca, sa, a, d = kin.ca[i], kin.sa[i], kin.a[i], kin.d[i]
        tx, ty, tz = a * c, a * s, d
        o = [o[r] + ((R[r][0] * tx + R[r][1] * ty) + R[r][2] * tz) for r in range(3)]
        m01 = -(s * ca)
        m02 = s * sa
        m11 = c * ca
        m12 = -(c * sa)
        R = [
            [
                R[r][0] * c + R[r][1] * s,
                (R[r][0] * m01 + R[r][1] * m11) + R[r][2] * sa,
                (R[r][0] * m02 + R[r][1] * m12) + R[r][2] * ca,
            ]
            for r in range(3)
        ]
    t = kin.tool
    pe = [o[r] + ((R[r][0] * t[0] + R[r][1] * t[1]) + R[r][2] * t[2]) for r in range(3)]
    return zs, os_, R, pe


def ur20_jacobian(q, kin: KinTable = UR20_KIN) -> np.ndarray:
    """6x6 geometric Jacobian (rows 0-2 linear, 3-5 angular; MPC:253) of the tip
    in the ``base_link`` frame.  Stands in for MoveIt's getJacobian (DYN:82)."""
    zs, os_, _, pe = _chain([float(v) for v in q], kin)
    J = np.empty((6, 6))
    for i in range(6):
        z0, z1, z2 = zs[i]
        d0, d1, d2 = pe[0] - os_[i][0], pe[1] - os_[i][1], pe[2] - os_[i][2]
        J[0, i] = z1 * d2 - z2 * d1
        J[1, i] = z2 * d0 - z0 * d2
        J[2, i] = z0 * d1 - z1 * d0
        J[3, i] = z0
        J[4, i] = z1
        J[5, i] = z2
    return J


def ur20_fk(q, kin: KinTable = UR20_KIN):
    """Tip position (3,) and roll/pitch/yaw (3,) in ``base_link``; rpy follows
    tf.transformations.euler_from_quaternion's default 'sxyz' convention, as
    used by the reference when it builds ``ee_pose`` (SRV:63-75)."""
    _, _, R, pe = _chain([float(v) for v in q], kin)
    cy = math.sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0])
    if cy > 1e-12:
        roll = math.atan2(R[2][1], R[2][2])
        pitch = math.atan2(-R[2][0], cy)
        yaw = math.atan2(R[1][0], R[0][0])
    else:
        roll = math.atan2(-R[1][2], R[1][1])
        pitch = math.atan2(-R[2][0], cy)
        yaw = 0.0
    return np.array(pe), np.array([roll, pitch, yaw])


# --------------------------------------------------------------------------
# Parameters (values = controller_params.yaml; code defaults are in Appendix A
# of SURVEY.md and are applied by the product's loader, not here)
# --------------------------------------------------------------------------

Q_LIM = 2.0 * np.pi  # DYN:31-38
V_LIM = (  # DYN:40-47
    2.094395102393195,
    2.094395102393195,
    2.617993877991494,
    3.665191429188092,
    3.665191429188092,
    3.665191429188092,
)


@dataclass
class RefParams:
    horizon: int = 1  # YAML:6
    dt: float = 0.07  # URMPC.dt (MPC:30) — used by predict_trajectory (MPC:705)
    dyn_dt: Optional[float] = None  # UR20Dynamics.dt (DYN:19) — used by _cost_function's rollout (MPC:686)
    w_position: float = 20.0
    w_orientation: float = 3.0
    w_control: float = 0.1
    w_smooth: float = 0.3
    w_damping: float = 2.0
    w_linear: float = 30.0
    w_angular: float = 30.0
    use_dynamic_model: bool = True
    base_mass: float = 65.0
    base_Iz: float = 8.0
    k_v: float = 25.0
    k_omega: float = 4.0
    tau_linear: float = 0.45
    tau_angular: float = 0.30
    fd_eps: float = 1e-3  # MPC:197
    ftol: float = 1e-3  # MPC:195
    maxiter: int = 50  # MPC:196
    kin: KinTable = field(default_factory=KinTable)

    def __post_init__(self):
        if self.dyn_dt is None:
            self.dyn_dt = self.dt
        if self.horizon <= 0:
            raise ValueError("Horizon must be positive")  # MPC:33-34
        if self.dt <= 0:
            raise ValueError("Time step must be positive")  # MPC:35-36


class OptimizationError(Exception):
    pass


class NumpyRefMPC:
    """Reference-shaped restatement of URMPC's hot path (one scenario at a time)."""

    n_q = 6
    n_controls = 6

    def __init__(self, params: RefParams, jacobian=None):
        self.p = params
        self.horizon = params.horizon
        self.dt = params.dt
        self.kin = params.kin
        self._jac = jacobian if jacobian is not None else (lambda q: ur20_jacobian(q, self.kin))
        self.q_lim = np.array([[-Q_LIM, Q_LIM]] * 6)
        self.v_lim = np.array(V_LIM)
        self.J_cached = None
        self.previous_control = None  # "hasattr(self,'previous_control')" in the reference (MPC:209,313)
        self.previous_solution = None
        self.n_jac_calls = 0
        # stateful filters of the kinematic predictor (MPC:532-572)
        self._prev_base_vel = None
        self._prev_accel = None
        self._prev_base_ang_vel = None
        self._prev_ang_accel = None

    # ---- a5 ---------------------------------------------------------------
    def get_jacobian(self, q) -> np.ndarray:
        self.n_jac_calls += 1
        return np.array(self._jac(np.asarray(q, dtype=float)))

    # ---- a9: MPC:714-720 --------------------------------------------------
    @staticmethod
    def world_to_base(v, yaw):
        c, s = np.cos(yaw), np.sin(yaw)
        rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
        return rot @ v

    # ---- a4: DYN:89-130 ---------------------------------------------------
    def predict_next_state(self, state, u, dt=None):
        if dt is None:
            dt = self.p.dyn_dt
        q = state[:6]
        q_new = q + u * dt
        twist = self.get_jacobian(q) @ u
        p_new = state[12:15] + twist[:3] * dt
        o_new = state[15:18] + twist[3:] * dt
        return np.concatenate([q_new, u, p_new, o_new])

    # ---- a7: MPC:694-710 --------------------------------------------------
    def predict_trajectory(self, state0, U, horizon):
        out = [state0]
        s = state0.copy()
        for k in range(horizon):
            s = self.predict_next_state(s, U[k], self.dt)
            out.append(s)
        return out

    def _compensation_targets(self, p_ee, base):
        """target linear / angular velocity of the arm in the base frame (MPC:263-283)."""
        lever = p_ee - base["position"]
        v_rot = np.cross(base["angular_velocity"], lever)
        tl_w = -(base["linear_velocity"][:3] + v_rot)
        ta_w = -base["angular_velocity"]
        yaw = base["orientation"][2]
        return self.world_to_base(tl_w, yaw), self.world_to_base(ta_w, yaw)

    # ---- a2: MPC:249-328 --------------------------------------------------
    def stage_cost(self, state, target, base, u, k):
        Jc = self.J_cached
        p_ee = state[12:15]
        o_ee = state[15:18]
        v_lin = Jc[:3, :] @ u
        v_ang = Jc[3:, :] @ u
        tl, ta = self._compensation_targets(p_ee, base)
        e_lin = v_lin - tl
        e_ang = v_ang - ta
        e_pos = p_ee - target["position"]
        e_ori = o_ee - target["orientation"]
        decay = 0.95 ** k
        P = self.p
        c_lin = P.w_linear * decay * np.sum(e_lin ** 2)
        c_ang = P.w_angular * decay * np.sum(e_ang ** 2)
        c_pos = P.w_position * decay * np.sum(e_pos ** 2)
        c_ori = P.w_orientation * decay * np.sum(e_ori ** 2)
        c_ctl = P.w_control * np.sum(u ** 2)
        if k > 0 and self.previous_control is not None:
            c_damp = P.w_damping * np.sum((u - self.previous_control) ** 2)
        else:
            c_damp = 0.0
        c_smooth = 0.0
        if k > 0 and self.horizon > 1:
            c_smooth = P.w_smooth * np.sum(u ** 2)
        return c_lin + c_ang + c_pos + c_ori + c_ctl + c_damp + c_smooth

    # ---- a3: MPC:330-381 --------------------------------------------------
    def terminal_cost(self, state, target, base):
        p_ee = state[12:15]
        o_ee = state[15:18]
        dq = state[6:12]
        JH = self.get_jacobian(state[:6])
        v_lin = JH[:3, :] @ dq
        v_ang = JH[3:, :] @ dq
        tl, ta = self._compensation_targets(p_ee, base)
        e_lin = v_lin - tl
        e_ang = v_ang - ta
        e_pos = p_ee - target["position"]
        e_ori = o_ee - target["orientation"]
        P = self.p
        factor = 2.0
        return (
            P.w_linear * factor * np.sum(e_lin ** 2)
            + P.w_angular * factor * np.sum(e_ang ** 2)
            + P.w_position * np.sum(e_pos ** 2)
            + P.w_orientation * np.sum(e_ori ** 2)
        )

    # ---- a1: MPC:665-692 --------------------------------------------------
    def cost_function(self, x, state0, target, base_traj):
        U = x.reshape(self.horizon, 6)
        total = 0.0
        s = state0.copy()
        for k in range(self.horizon):
            total += self.stage_cost(s, target, base_traj[k], U[k], k)
            s = self.predict_next_state(s, U[k])
        total += self.terminal_cost(s, target, base_traj[-1])
        return total

    # ---- a6: MPC:383-418 --------------------------------------------------
    def constraint(self, x, state0):
        U = x.reshape(self.horizon, 6)
        rows = []
        for s in self.predict_trajectory(state0, U, self.horizon):
            q = s[:6]
            dq = s[6:12]
            rows.extend(np.minimum(q - self.q_lim[:, 0], self.q_lim[:, 1] - q))
            rows.extend(self.v_lim - np.abs(dq))
        return np.array(rows)

    def constraint_joints_only(self, x, state0):
        """Same values as ``constraint`` without the unused EE integration
        (the H Jacobian calls the reference wastes)."""
        U = x.reshape(self.horizon, 6)
        q = state0[:6].copy()
        dq = state0[6:12].copy()
        rows = []
        for k in range(self.horizon + 1):
            rows.extend(np.minimum(q - self.q_lim[:, 0], self.q_lim[:, 1] - q))
            rows.extend(self.v_lim - np.abs(dq))
            if k < self.horizon:
                q = q + U[k] * self.dt
                dq = U[k]
        return np.array(rows)

    # ---- a8: MPC:420-430 --------------------------------------------------
    def bounds(self):
        return [(-self.v_lim[j], self.v_lim[j]) for _ in range(self.horizon) for j in range(6)]

    # ---- a10: scipy's own FD, as SLSQP wires it (_slsqp_py.py:299-313,381-387) ----
    def fd_eval(self, x, state0, target, base_traj, fast_constraint=True):
        """One "eval": f, c, grad f, dc/dx exactly as scipy SLSQP would request them."""
        from scipy.optimize._numdiff import approx_derivative

        lb = np.array([b[0] for b in self.bounds()])
        ub = np.array([b[1] for b in self.bounds()])
        fun = lambda z: self.cost_function(z, state0, target, base_traj)
        con = (lambda z: self.constraint_joints_only(z, state0)) if fast_constraint else (
            lambda z: self.constraint(z, state0))
        f0 = fun(x)
        c0 = con(x)
        g = approx_derivative(fun, x, method="2-point", abs_step=self.p.fd_eps, f0=f0, bounds=(lb, ub))
        a = approx_derivative(con, x, method="2-point", abs_step=self.p.fd_eps, f0=c0, bounds=(lb, ub))
        return f0, c0, np.atleast_1d(g), np.atleast_2d(a)

    # ---- a12: MPC:606-651 -------------------------------------------------
    def predict_base_dynamic(self, base_state, horizon):
        P = self.p
        dt = self.dt
        pos = base_state["position"].copy()
        yaw = base_state["orientation"][2]
        v = base_state["linear_velocity"][0]
        w = base_state["angular_velocity"][2]
        v_cmd = base_state.get("v_cmd", v)
        w_cmd = base_state.get("omega_cmd", w)
        out = []
        for _ in range(horizon):
            force = (P.base_mass / P.tau_linear) * (v_cmd - v)
            acc = (force - P.k_v * v) / P.base_mass
            v = v + acc * dt
            torque = (P.base_Iz / P.tau_angular) * (w_cmd - w)
            alpha = (torque - P.k_omega * w) / P.base_Iz
            w = w + alpha * dt
            yaw = yaw + w * dt
            c, s = np.cos(yaw), np.sin(yaw)
            vx, vy = v * c, v * s
            pos = pos + np.array([vx, vy, 0.0]) * dt
            out.append({
                "position": pos.copy(),
                "orientation": np.array([0.0, 0.0, yaw]),
                "linear_velocity": np.array([vx, vy, 0.0]),
                "angular_velocity": np.array([0.0, 0.0, w]),
            })
        return out

    # ---- a12: MPC:527-601 (stateful) --------------------------------------
    def predict_base_kinematic(self, base_state, horizon):
        dt = self.dt
        lin = base_state["linear_velocity"]
        ang = base_state["angular_velocity"]
        if self._prev_base_vel is not None:
            raw = (lin - self._prev_base_vel) / dt
            if self._prev_accel is not None:
                acc = 0.5 * raw + (1 - 0.5) * self._prev_accel
                acc = np.clip(acc, -0.5, 0.5)
            else:
                acc = raw * 0.5
            self._prev_accel = acc.copy()
        else:
            acc = np.zeros(3)
        self._prev_base_vel = lin.copy()
        if self._prev_base_ang_vel is not None:
            raw = (ang - self._prev_base_ang_vel) / dt
            if self._prev_ang_accel is not None:
                aacc = 0.9 * raw + (1 - 0.9) * self._prev_ang_accel
                aacc = np.clip(aacc, -0.3, 0.3)
            else:
                aacc = raw * 0.5
            self._prev_ang_accel = aacc.copy()
        else:
            aacc = np.zeros(3)
        self._prev_base_ang_vel = ang.copy()
        out = []
        for k in range(horizon):
            decay = min(1.0, max(0.3, 1.0 - 0.2 * k))
            out.append({
                "position": (base_state["position"] + k * dt * lin + 0.5 * k * dt ** 2 * acc * decay).copy(),
                "orientation": (base_state["orientation"] + k * dt * ang + 0.5 * k * dt ** 2 * aacc * decay).copy(),
                "linear_velocity": (lin + k * dt * acc * decay).copy(),
                "angular_velocity": (ang + k * dt * aacc * decay).copy(),
            })
        return out

    def predict_base_motion(self, base_state, horizon):
        if self.p.use_dynamic_model:
            return self.predict_base_dynamic(base_state, horizon)
        return self.predict_base_kinematic(base_state, horizon)

    # ---- a13: MPC:83-247 --------------------------------------------------
    def dls_guess(self, Jc, base_state, ee_pose):
        lever = ee_pose["position"] - base_state["position"]
        v_ind = base_state["linear_velocity"] + np.cross(base_state["angular_velocity"], lever)
        yaw = base_state["orientation"][2]
        tl = self.world_to_base(-v_ind, yaw)
        ta = self.world_to_base(-base_state["angular_velocity"], yaw)
        Jf = np.vstack([Jc[:3, :], Jc[3:, :]])
        twist = np.concatenate([tl, ta])
        pinv = Jf.T @ np.linalg.inv(Jf @ Jf.T + 0.01 * np.eye(6))
        return pinv @ twist

    def initial_x(self, u_init):
        if self.previous_solution is not None:
            x0 = np.roll(self.previous_solution, -6)
            x0[-6:] = u_init
        else:
            x0 = np.zeros(6 * self.horizon)
            for i in range(self.horizon):
                x0[6 * i:6 * i + 6] = u_init
        return x0

    def compute_control(self, joint_state, ee_pose, target, base_state, info=None):
        from scipy.optimize import minimize

        u_init = None
        try:
            if (np.linalg.norm(base_state["linear_velocity"]) < 0.01
                    and np.linalg.norm(base_state["angular_velocity"]) < 0.01):
                return np.zeros(6)
            state0 = np.concatenate([joint_state["position"], joint_state["velocity"],
                                     ee_pose["position"], ee_pose["orientation"]])
            base_traj = self.predict_base_motion(base_state, self.horizon)
            self.J_cached = self.get_jacobian(joint_state["position"]).copy()
            u_init = self.dls_guess(self.J_cached, base_state, ee_pose)
            x0 = self.initial_x(u_init)
            res = minimize(
                fun=lambda z: self.cost_function(z, state0, target, base_traj),
                x0=x0, method="SLSQP", bounds=self.bounds(),
                constraints=[{"type": "ineq", "fun": lambda z: self.constraint(z, state0)}],
                options={"ftol": self.p.ftol, "maxiter": self.p.maxiter, "eps": self.p.fd_eps, "disp": False},
            )
            if info is not None:
                info.update(x=res.x.copy(), nit=res.nit, fun=float(res.fun), status=res.status,
                            nfev=res.nfev, x0=x0.copy())
            if not res.success:
                raise OptimizationError(res.message)
            u = res.x[:6]
            if self.previous_control is not None:
                u = 0.5 * u + (1.0 - 0.5) * self.previous_control
            self.previous_control = u.copy()
            self.previous_solution = res.x
            return np.clip(u, -0.5, 0.5)
        except OptimizationError:
            return u_init
        except ValueError:
            return np.zeros(6)


def base_traj_to_array(base_traj: List[Dict]) -> np.ndarray:
    """list of H dicts -> (H, 12) rows [P(3), Theta(3), V(3), Omega(3)]."""
    return np.array([np.concatenate([b["position"], b["orientation"], b["linear_velocity"][:3],
                                     b["angular_velocity"]]) for b in base_traj])


def base_traj_from_array(arr: np.ndarray) -> List[Dict]:
    return [{"position": r[0:3].copy(), "orientation": r[3:6].copy(), "linear_velocity": r[6:9].copy(),
             "angular_velocity": r[9:12].copy()} for r in np.asarray(arr).reshape(-1, 12)]
