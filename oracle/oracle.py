"""TEST INFRASTRUCTURE — ctypes wrapper around the C oracle (oracle/mpc_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from oracle.numpy_ref import Q_LIM, V_LIM, KinTable, RefParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpc_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("horizon", C.c_int32), ("reserved", C.c_int32),
        ("dt_cost", C.c_double), ("dt_con", C.c_double),
        ("w_position", C.c_double), ("w_orientation", C.c_double), ("w_control", C.c_double),
        ("w_smooth", C.c_double), ("w_damping", C.c_double), ("w_linear", C.c_double),
        ("w_angular", C.c_double),
        ("terminal_factor", C.c_double), ("fd_eps", C.c_double),
        ("q_lo", C.c_double * 6), ("q_hi", C.c_double * 6), ("v_lim", C.c_double * 6),
        ("dh_a", C.c_double * 6), ("dh_d", C.c_double * 6), ("dh_ca", C.c_double * 6),
        ("dh_sa", C.c_double * 6),
        ("base_c", C.c_double), ("base_s", C.c_double), ("tool", C.c_double * 3),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mpc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        pp = C.POINTER(OrcParams)
        L.orc_jacobian.argtypes = [pp, dp, dp]
        L.orc_fk.argtypes = [pp, dp, dp, dp]
        L.orc_cost.restype = C.c_double
        L.orc_cost.argtypes = [pp, dp, dp, dp, dp, dp, dp, dp, C.c_int]
        L.orc_constraint.argtypes = [pp, dp, dp, dp]
        L.orc_fd_step.restype = C.c_double
        L.orc_fd_step.argtypes = [C.c_double] * 4
        L.orc_eval_fd_batch.argtypes = [pp, dp, C.c_int] + [dp] * 6 + [C.POINTER(C.c_uint8)] + [dp] * 4
        L.orc_eval_batch.argtypes = [pp, dp, C.c_int] + [dp] * 6 + [C.POINTER(C.c_uint8)] + [dp] * 2
        L.orc_predict_base_dynamic.argtypes = [dp, dp, C.c_double, C.c_int, dp]
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def make_params(p: RefParams) -> OrcParams:
    k: KinTable = p.kin
    o = OrcParams()
    o.horizon = p.horizon
    o.dt_cost = p.dyn_dt
    o.dt_con = p.dt
    o.w_position, o.w_orientation, o.w_control = p.w_position, p.w_orientation, p.w_control
    o.w_smooth, o.w_damping, o.w_linear, o.w_angular = p.w_smooth, p.w_damping, p.w_linear, p.w_angular
    o.terminal_factor = 2.0
    o.fd_eps = p.fd_eps
    for j in range(6):
        o.q_lo[j], o.q_hi[j], o.v_lim[j] = -Q_LIM, Q_LIM, V_LIM[j]
        o.dh_a[j], o.dh_d[j], o.dh_ca[j], o.dh_sa[j] = k.a[j], k.d[j], k.ca[j], k.sa[j]
    o.base_c, o.base_s = k.base_c, k.base_s
    for j in range(3):
        o.tool[j] = k.tool[j]
    return o


def decay_table(H: int) -> np.ndarray:
    """0.95**k exactly as CPython evaluates MPC:294."""
    return np.array([0.95 ** k for k in range(H)], dtype=np.float64)


class COracle:
    """Batched C oracle: arrays are [N][...] C-contiguous float64."""

    def __init__(self, p: RefParams):
        self.p = p
        self.cp = make_params(p)
        self.H = p.horizon
        self.n = 6 * self.H
        self.m = 12 * (self.H + 1)
        self.decay = decay_table(self.H)

    def jacobian(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 6)
        J = np.empty((q.shape[0], 6, 6))
        for i in range(q.shape[0]):
            lib().orc_jacobian(C.byref(self.cp), _dp(q[i]), _dp(J[i]))
        return J

    def fk(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 6)
        pos = np.empty((q.shape[0], 3))
        rpy = np.empty((q.shape[0], 3))
        for i in range(q.shape[0]):
            lib().orc_fk(C.byref(self.cp), _dp(q[i]), _dp(pos[i]), _dp(rpy[i]))
        return pos, rpy

    @staticmethod
    def _prep(sc):
        out = {}
        for k in ("x", "state0", "target", "base_traj", "Jc", "u_prev"):
            v = sc.get(k)
            out[k] = None if v is None else np.ascontiguousarray(v, dtype=np.float64)
        hp = sc.get("has_prev")
        out["has_prev"] = None if hp is None else np.ascontiguousarray(hp, dtype=np.uint8)
        if out["u_prev"] is None:
            out["has_prev"] = None
        return out

    def eval_fd(self, sc, want_jac=True, want_grad=True):
        a = self._prep(sc)
        N = a["x"].reshape(-1, self.n).shape[0]
        cost = np.empty(N)
        grad = np.empty((N, self.n)) if want_grad else None
        con = np.empty((N, self.m))
        jac = np.empty((N, self.n, self.m)) if want_jac else None
        hp = None if a["has_prev"] is None else a["has_prev"].ctypes.data_as(C.POINTER(C.c_uint8))
        lib().orc_eval_fd_batch(C.byref(self.cp), _dp(self.decay), N, _dp(a["x"]), _dp(a["state0"]),
                                _dp(a["target"]), _dp(a["base_traj"]), _dp(a["Jc"]), _dp(a["u_prev"]), hp,
                                _dp(cost), _dp(grad), _dp(con), _dp(jac))
        return cost, grad, con, jac

    def eval(self, sc):
        a = self._prep(sc)
        N = a["x"].reshape(-1, self.n).shape[0]
        cost = np.empty(N)
        con = np.empty((N, self.m))
        hp = None if a["has_prev"] is None else a["has_prev"].ctypes.data_as(C.POINTER(C.c_uint8))
        lib().orc_eval_batch(C.byref(self.cp), _dp(self.decay), N, _dp(a["x"]), _dp(a["state0"]),
                             _dp(a["target"]), _dp(a["base_traj"]), _dp(a["Jc"]), _dp(a["u_prev"]), hp,
                             _dp(cost), _dp(con))
        return cost, con

    def predict_base_dynamic(self, base12):
        base12 = np.ascontiguousarray(base12, dtype=np.float64).reshape(-1, 12)
        p = self.p
        dyn = np.array([p.base_mass, p.base_Iz, p.k_v, p.k_omega, p.tau_linear, p.tau_angular])
        out = np.empty((base12.shape[0], self.H, 12))
        for i in range(base12.shape[0]):
            lib().orc_predict_base_dynamic(_dp(base12[i]), _dp(dyn), p.dt, self.H, _dp(out[i]))
        return out


class OracleKinematics:
    """``kin`` backend for mpc_amd.scenarios.make_scenarios backed by the C oracle (tests only)."""

    def __init__(self, co: COracle):
        self.co = co

    def fk(self, q):
        pos, rpy = self.co.fk(q)
        return np.concatenate([pos, rpy], axis=1)

    def jacobian(self, q):
        return self.co.jacobian(q)

    def predict_base_dynamic(self, base):
        return self.co.predict_base_dynamic(base)


def ref_params_from(mp) -> RefParams:
    """RefParams (oracle) with the values of an mpc_amd MpcParams (duck-typed, tests only)."""
    kin = KinTable(a=tuple(mp.kin.a), d=tuple(mp.kin.d), ca=tuple(mp.kin.ca), sa=tuple(mp.kin.sa),
                   base_c=mp.kin.base_c, base_s=mp.kin.base_s, tool=tuple(mp.kin.tool))
    return RefParams(horizon=mp.horizon, dt=mp.dt, dyn_dt=mp.dyn_dt, w_position=mp.w_position,
                     w_orientation=mp.w_orientation, w_control=mp.w_control, w_smooth=mp.w_smooth,
                     w_damping=mp.w_damping, w_linear=mp.w_linear, w_angular=mp.w_angular,
                     use_dynamic_model=mp.use_dynamic_model, base_mass=mp.base_mass, base_Iz=mp.base_Iz,
                     k_v=mp.k_v, k_omega=mp.k_omega, tau_linear=mp.tau_linear, tau_angular=mp.tau_angular,
                     fd_eps=mp.fd_eps, ftol=mp.ftol, maxiter=mp.maxiter, kin=kin)
