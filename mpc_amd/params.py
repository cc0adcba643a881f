"""Parameters of the MPC inner loop and their loader.

Resolution rule = the reference's ``rospy.get_param('~mpc_controller/<key>', <code default>)``:
the value in ``controller_params.yaml`` if the key is present, else the default written in the
reference's constructor (ur_mpc.py:29-81, ur20_dynamics.py:19)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field, replace
from typing import Any, Dict, Optional, Sequence

import numpy as np

from . import _ffi

JOINT_NAMES = ("shoulder_pan_joint", "shoulder_lift_joint", "elbow_joint",
               "wrist_1_joint", "wrist_2_joint", "wrist_3_joint")  # ur20_dynamics.py:40-47


@dataclass(frozen=True)
class KinematicTable:
    """Standard-DH table of the arm plus the fixed transforms either side of it.

    The reference obtains its Jacobian from MoveIt on a URDF that is not part of its tree
    (ur20_dynamics.py:82); this table (nominal UR20 DH, ``base_link`` rotated pi about z from
    the DH base, 0.02 m adapter offset along tool z from mpc_robot.xacro:65-70) is this
    package's stand-in.  Twists are stored as exact cos/sin."""

    a: Sequence[float] = (0.0, -0.8620, -0.7287, 0.0, 0.0, 0.0)
    d: Sequence[float] = (0.2363, 0.0, 0.0, 0.2010, 0.1593, 0.1543)
    ca: Sequence[float] = (0.0, 1.0, 1.0, 0.0, 0.0, 1.0)
    sa: Sequence[float] = (1.0, 0.0, 0.0, 1.0, -1.0, 0.0)
    base_c: float = -1.0
    base_s: float = 0.0
    tool: Sequence[float] = (0.0, 0.0, 0.02)


# code defaults of the reference constructors (used when the YAML lacks the key)
_CODE_DEFAULTS: Dict[str, Any] = {
    "horizon": 10, "dt": 0.1, "dyn_dt": 0.05,
    "weights/position": 3.0, "weights/orientation": 2.0, "weights/control": 0.005,
    "weights/smooth": 0.02, "weights/damping": 0.5,
    "active_weights/linear": 50.0, "active_weights/angular": 5.0,
    "base_dynamics/use_dynamic_model": False, "base_dynamics/mass": 180.0, "base_dynamics/I_z": 62.0,
    "base_dynamics/k_v": 25.0, "base_dynamics/k_omega": 8.0,
    "base_dynamics/tau_linear": 0.45, "base_dynamics/tau_angular": 0.30,
}


@dataclass(frozen=True)
class MpcParams:
    horizon: int = 10                 # ur_mpc.py:29
    dt: float = 0.1                   # URMPC.dt, ur_mpc.py:30
    dyn_dt: float = 0.05              # UR20Dynamics.dt, ur20_dynamics.py:19
    w_position: float = 3.0
    w_orientation: float = 2.0
    w_control: float = 0.005
    w_smooth: float = 0.02
    w_damping: float = 0.5
    w_linear: float = 50.0
    w_angular: float = 5.0
    use_dynamic_model: bool = False
    base_mass: float = 180.0
    base_Iz: float = 62.0
    k_v: float = 25.0
    k_omega: float = 8.0
    tau_linear: float = 0.45
    tau_angular: float = 0.30
    # hard-coded constants of the path
    terminal_factor: float = 2.0      # ur_mpc.py:373
    decay_base: float = 0.95          # ur_mpc.py:294
    fd_eps: float = 1e-3              # ur_mpc.py:197
    ftol: float = 1e-3                # ur_mpc.py:195
    maxiter: int = 50                 # ur_mpc.py:196
    q_limit: float = 2.0 * math.pi    # ur20_dynamics.py:31-38
    v_lim: Sequence[float] = (2.094395102393195, 2.094395102393195, 2.617993877991494,
                              3.665191429188092, 3.665191429188092, 3.665191429188092)
    kin: KinematicTable = field(default_factory=KinematicTable)
    algo: int = _ffi.ALGO_AUTO

    def __post_init__(self):
        if self.horizon <= 0:
            raise ValueError("Horizon must be positive")       # ur_mpc.py:33-34
        if self.dt <= 0 or self.dyn_dt <= 0:
            raise ValueError("Time step must be positive")     # ur_mpc.py:35-36
        if self.horizon > _ffi.MAX_HORIZON:
            raise ValueError(f"horizon {self.horizon} > {_ffi.MAX_HORIZON}")

    # ---- sizes -----------------------------------------------------------------------------
    @property
    def n(self) -> int:
        return 6 * self.horizon

    @property
    def m(self) -> int:
        return 12 * (self.horizon + 1)

    def replace(self, **kw) -> "MpcParams":
        return replace(self, **kw)

    # ---- loaders ---------------------------------------------------------------------------
    @classmethod
    def from_param_tree(cls, tree: Optional[Dict[str, Any]], **overrides) -> "MpcParams":
        """tree = the dict under the ``mpc_controller`` key of controller_params.yaml (or None)."""
        tree = tree or {}

        def get(key):
            node = tree
            for part in key.split("/"):
                if isinstance(node, dict) and part in node:
                    node = node[part]
                else:
                    # UR20Dynamics and URMPC read the same 'dt' key with different defaults
                    return _CODE_DEFAULTS[key]
            return node

        dt_present = isinstance(tree, dict) and "dt" in tree
        kw = dict(
            horizon=int(get("horizon")), dt=float(get("dt")),
            dyn_dt=float(tree["dt"]) if dt_present else _CODE_DEFAULTS["dyn_dt"],
            w_position=float(get("weights/position")), w_orientation=float(get("weights/orientation")),
            w_control=float(get("weights/control")), w_smooth=float(get("weights/smooth")),
            w_damping=float(get("weights/damping")),
            w_linear=float(get("active_weights/linear")), w_angular=float(get("active_weights/angular")),
            use_dynamic_model=bool(get("base_dynamics/use_dynamic_model")),
            base_mass=float(get("base_dynamics/mass")), base_Iz=float(get("base_dynamics/I_z")),
            k_v=float(get("base_dynamics/k_v")), k_omega=float(get("base_dynamics/k_omega")),
            tau_linear=float(get("base_dynamics/tau_linear")), tau_angular=float(get("base_dynamics/tau_angular")),
        )
        kw.update(overrides)
        return cls(**kw)

    @classmethod
    def from_yaml(cls, path: str, **overrides) -> "MpcParams":
        import yaml

        with open(path) as fh:
            doc = yaml.safe_load(fh) or {}
        return cls.from_param_tree(doc.get("mpc_controller", {}), **overrides)

    @classmethod
    def reference_yaml_defaults(cls, **overrides) -> "MpcParams":
        """The values of the reference's shipped controller_params.yaml (config/controller_params.yaml:6-45)."""
        kw = dict(horizon=1, dt=0.07, dyn_dt=0.07, w_position=20.0, w_orientation=3.0, w_control=0.1,
                  w_smooth=0.3, w_damping=2.0, w_linear=30.0, w_angular=30.0, use_dynamic_model=True,
                  base_mass=65.0, base_Iz=8.0, k_v=25.0, k_omega=4.0, tau_linear=0.45, tau_angular=0.30)
        kw.update(overrides)
        return cls(**kw)

    # ---- views for the C-ABI ---------------------------------------------------------------
    def decay_table(self) -> np.ndarray:
        """decay[k] = 0.95**k by CPython's float pow, bit-identical to ur_mpc.py:294."""
        return np.array([self.decay_base ** k for k in range(self.horizon)], dtype=np.float64)

    def bounds(self):
        """ur_mpc.py:420-430"""
        return [(-self.v_lim[j], self.v_lim[j]) for _ in range(self.horizon) for j in range(6)]

    def base_dyn_array(self) -> np.ndarray:
        return np.array([self.base_mass, self.base_Iz, self.k_v, self.k_omega, self.tau_linear,
                         self.tau_angular], dtype=np.float64)

    def to_c(self) -> _ffi.MpcParamsC:
        c = _ffi.MpcParamsC()
        c.horizon, c.algo = self.horizon, self.algo
        c.dt_cost, c.dt_con = self.dyn_dt, self.dt
        c.w_position, c.w_orientation, c.w_control = self.w_position, self.w_orientation, self.w_control
        c.w_smooth, c.w_damping = self.w_smooth, self.w_damping
        c.w_linear, c.w_angular = self.w_linear, self.w_angular
        c.terminal_factor, c.fd_eps = self.terminal_factor, self.fd_eps
        for j in range(6):
            c.q_lo[j], c.q_hi[j], c.v_lim[j] = -self.q_limit, self.q_limit, self.v_lim[j]
            c.dh_a[j], c.dh_d[j] = self.kin.a[j], self.kin.d[j]
            c.dh_ca[j], c.dh_sa[j] = self.kin.ca[j], self.kin.sa[j]
        c.base_c, c.base_s = self.kin.base_c, self.kin.base_s
        for j in range(3):
            c.tool[j] = self.kin.tool[j]
        return c
