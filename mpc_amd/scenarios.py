"""Seeded synthetic scenario family of the benchmark.

A scenario = one MPC tick on a platform that moves as ``A sin(w t + phi)`` (the motion driven by
the reference's tests/simulate_platform_motion.py:17,23,42-43,59: f = 0.1 Hz, A = 0.2 m on the
prismatic x axis or 20 deg about z), the arm near one of the two SRDF start poses
(mpc_moveit_config/config/mpc.srdf:16-31) and a goal pose near the current tip pose.

All kinematics (tip pose, Jacobian, platform prediction) come from a ``kin`` backend with
``fk(q)->[N,6]``, ``jacobian(q)->[N,6,6]``, ``predict_base_dynamic(base)->[N,H,12]`` returning numpy
arrays: the product passes :class:`DeviceKinematics` (libmpc_amd), tests may pass the CPU oracle.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .params import MpcParams

# SRDF group states in joint order pan, lift, elbow, wrist1, wrist2, wrist3
Q_LINEAR_READY = np.array([5.1508, -0.7884, -2.2445, -0.9465, 1.2523, 3.4515])
Q_ANGULAR_READY = np.array([4.9015, -1.3217, -2.0265, -0.5868, 1.4311, 3.2573])
ARM_OFFSET = np.array([0.0, 0.1, 0.5])       # base_joint origin, mpc_robot.xacro:17-22
PLATFORM_FREQ_HZ = 0.1                        # simulate_platform_motion.py:42
BASE_SEED = 20250614


@dataclass
class HostScenarios:
    """numpy (host) copy of a scenario batch, rows [N][...]."""

    motion: str
    q0: np.ndarray          # [N,6]
    base: np.ndarray        # [N,12] platform state [P, Theta, V, Omega]
    state0: np.ndarray      # [N,18]
    target: np.ndarray      # [N,6]
    base_traj: np.ndarray   # [N,H,12]
    Jc: np.ndarray          # [N,36]
    x: np.ndarray           # [N,6H] evaluation point / SLSQP start
    u_init: np.ndarray      # [N,6]  DLS guess (ur_mpc.py:173-175)
    u_prev: Optional[np.ndarray]    # [N,6] or None
    has_prev: Optional[np.ndarray]  # [N] uint8 or None
    active: np.ndarray      # [N] bool: False where compute_control exits early (ur_mpc.py:108)

    @property
    def count(self):
        return self.q0.shape[0]

    def as_oracle_dict(self, lo=0, hi=None):
        sl = slice(lo, hi)
        return dict(x=self.x[sl], state0=self.state0[sl], target=self.target[sl], base_traj=self.base_traj[sl],
                    Jc=self.Jc[sl], u_prev=None if self.u_prev is None else self.u_prev[sl],
                    has_prev=None if self.has_prev is None else self.has_prev[sl])


def world_to_base(v: np.ndarray, yaw: np.ndarray) -> np.ndarray:
    """Batched ur_mpc.py:714-720: the 3x3 rotation matrix times the vector, as a stacked matrix-vector
    product (numpy runs the same BLAS routine per row as the reference's ``R @ v``: identical bits)."""
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.zeros((v.shape[0], 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, s, -s, c, 1.0
    return (R @ v[:, :, None])[:, :, 0]


def dls_guess(Jc: np.ndarray, base: np.ndarray, ee_pos: np.ndarray) -> np.ndarray:
    """Batched damped-least-squares start of compute_control (ur_mpc.py:112-143,168-175)."""
    P, Th, V, Om = base[:, 0:3], base[:, 3:6], base[:, 6:9], base[:, 9:12]
    v_ind = V + np.cross(Om, ee_pos - P)
    tl = world_to_base(-v_ind, Th[:, 2])
    ta = world_to_base(-Om, Th[:, 2])
    twist = np.concatenate([tl, ta], axis=1)
    J = Jc.reshape(-1, 6, 6)
    Jt = np.transpose(J, (0, 2, 1))
    # the reference's operation order, (J^T inv(J J^T + 0.01 I)) twist, so that start points round as its do
    pinv = Jt @ np.linalg.inv(J @ Jt + 0.01 * np.eye(6))
    return (pinv @ twist[:, :, None])[:, :, 0]


def platform_state(motion: str, amplitude: np.ndarray, phase: np.ndarray, t: float = 0.0) -> np.ndarray:
    """Platform state [P, Theta, V, Omega] at time t of the motion the reference's simulator drives
    (tests/simulate_platform_motion.py:42-59): velocity command A w cos(w t + phi) on the prismatic x
    axis ('linear', A in m) or about z ('angular', A in rad)."""
    amplitude = np.atleast_1d(np.asarray(amplitude, float))
    phase = np.atleast_1d(np.asarray(phase, float))
    w = 2.0 * np.pi * PLATFORM_FREQ_HZ
    base = np.zeros((amplitude.shape[0], 12))
    pos, vel = amplitude * np.sin(w * t + phase), amplitude * w * np.cos(w * t + phase)
    if motion == "linear":
        base[:, 0], base[:, 6] = pos, vel
    elif motion == "angular":
        base[:, 5], base[:, 11] = pos, vel
    else:
        raise ValueError("motion must be 'linear' or 'angular'")
    return base


def world_ee_pose(pose_local: np.ndarray, base: np.ndarray) -> np.ndarray:
    """Tip pose in the world: platform pose o arm mount (mpc_robot.xacro:17-22) o FK in base_link."""
    yaw = base[:, 5]
    local = ARM_OFFSET + pose_local[:, 0:3]
    c, s = np.cos(yaw), np.sin(yaw)
    out = np.empty((pose_local.shape[0], 6))
    out[:, 0] = base[:, 0] + c * local[:, 0] - s * local[:, 1]
    out[:, 1] = base[:, 1] + s * local[:, 0] + c * local[:, 1]
    out[:, 2] = base[:, 2] + local[:, 2]
    out[:, 3:6] = pose_local[:, 3:6]
    out[:, 5] += yaw
    return out


def make_scenarios(params: MpcParams, kin, count: int, motion: str = "linear", cfg_id: int = 2,
                   with_prev: bool = False, goal_jitter: float = 0.05, first: int = 0) -> HostScenarios:
    """Scenarios ``first .. first+count-1`` of configuration ``cfg_id``.  Row i depends only on
    (cfg_id, i), so a shard [first, first+count) equals the same rows of the full batch."""
    if motion not in ("linear", "angular"):
        raise ValueError("motion must be 'linear' or 'angular'")
    H = params.horizon
    K = 14 + 6 * H
    rng = np.random.default_rng(BASE_SEED + cfg_id)
    if first:
        # Generator.uniform draws one double per value: skip the rows of earlier shards
        rng.bit_generator.advance(first * K)
    U = rng.random((count, K))
    q0 = (Q_LINEAR_READY if motion == "linear" else Q_ANGULAR_READY) + (U[:, 0:6] * 0.2 - 0.1)
    phi = U[:, 7] * 2.0 * np.pi
    A = (0.05 + U[:, 6] * 0.25) if motion == "linear" else np.radians(5.0 + U[:, 6] * 25.0)
    base = platform_state(motion, A, phi)
    ee0 = world_ee_pose(np.asarray(kin.fk(q0)), base)      # FK gives the tip in base_link
    p0, o0 = ee0[:, 0:3], ee0[:, 3:6]
    state0 = np.concatenate([q0, np.zeros((count, 6)), p0, o0], axis=1)   # dq0 = 0 (mpc_action_server.py:83)
    target = np.concatenate([p0, o0], axis=1) + (U[:, 8:14] * 2.0 - 1.0) * goal_jitter
    base_traj = np.asarray(kin.predict_base_dynamic(base))
    Jc = np.asarray(kin.jacobian(q0)).reshape(count, 36)
    u_init = dls_guess(Jc, base, p0)
    x = np.tile(u_init, (1, H)) + (U[:, 14:] * 0.1 - 0.05)
    lim = np.tile(np.asarray(params.v_lim), H)
    x = np.clip(x, -lim, lim)
    active = ~((np.linalg.norm(base[:, 6:9], axis=1) < 0.01) & (np.linalg.norm(base[:, 9:12], axis=1) < 0.01))
    u_prev = x[:, :6].copy() if with_prev else None
    has_prev = np.ones(count, dtype=np.uint8) if with_prev else None
    hs = HostScenarios(motion, q0, base, state0, target, base_traj, Jc, np.ascontiguousarray(x), u_init,
                       u_prev, has_prev, active)
    hs.amplitude, hs.phase = A, phi
    return hs


SWEEP_GRID = 32            # amplitude x phase lattice, 32 x 32 cells
SWEEP_GOALS = 1024         # goal poses per lattice cell
SWEEP_ROWS = SWEEP_GRID * SWEEP_GRID * SWEEP_GOALS     # 1,048,576 = BASELINE configs[4]
SWEEP_CFG_ID = 5


def sweep_goal_table(horizon: int, goal_jitter: float = 0.05) -> np.ndarray:
    """The seeded table of the sweep's 1,024 goal poses, [1024, 6 + 6H]: columns 0..5 the offset of the goal pose
    from the tip pose (U(-goal_jitter, goal_jitter), mpc_action_server.py:77 is the nominal zero offset), the
    rest the offset of the evaluation point / SLSQP start from the damped-least-squares guess (U(-0.05, 0.05))."""
    rng = np.random.default_rng(BASE_SEED + SWEEP_CFG_ID)
    U = rng.random((SWEEP_GOALS, 6 + 6 * horizon))
    U[:, :6] = (U[:, :6] * 2.0 - 1.0) * goal_jitter
    U[:, 6:] = U[:, 6:] * 0.1 - 0.05
    return U


def sweep_lattice(motion: str):
    """Cell centres of the 32 x 32 amplitude x phase lattice over the ranges of SURVEY.md §8(d): A in 0.05..0.30 m
    (prismatic platform joint; nominal 0.2, simulate_platform_motion.py:17) or 5..30 deg (revolute; nominal 20,
    :23), phi in 0..2 pi."""
    c = (np.arange(SWEEP_GRID) + 0.5) / SWEEP_GRID
    A = (0.05 + 0.25 * c) if motion == "linear" else np.radians(5.0 + 25.0 * c)
    return A, 2.0 * np.pi * c


def sweep_index(rows: np.ndarray):
    """Global sweep row -> (amplitude index, phase index, goal index): row = (ia * 32 + ip) * 1024 + goal."""
    rows = np.asarray(rows, dtype=np.int64)
    cell, goal = rows // SWEEP_GOALS, rows % SWEEP_GOALS
    return cell // SWEEP_GRID, cell % SWEEP_GRID, goal


def make_sweep(params: MpcParams, kin, first: int, count: int, motion: str = "linear",
               with_prev: bool = False, goal_jitter: float = 0.05) -> HostScenarios:
    """Rows ``first .. first+count-1`` of BASELINE configs[4]: the product of a 32 x 32 lattice of platform
    amplitude x phase (the motion family of tests/simulate_platform_motion.py:17,23,42-59) with 1,024 goal poses,
    1,048,576 rows in all, the arm at the SRDF start pose of the motion (mpc.srdf:16-31).  Global row i is lattice
    cell i // 1024 (amplitude index slowest) and goal pose i % 1024, whatever shard it is made in: rank r of an
    8-GPU run makes rows [131072 r, 131072 (r+1))."""
    if motion not in ("linear", "angular"):
        raise ValueError("motion must be 'linear' or 'angular'")
    if first < 0 or count < 0 or first + count > SWEEP_ROWS:
        raise ValueError(f"the sweep has rows 0..{SWEEP_ROWS - 1}")
    H = params.horizon
    ia, ip, goal = sweep_index(first + np.arange(count))
    A_axis, phi_axis = sweep_lattice(motion)
    A, phi = A_axis[ia], phi_axis[ip]
    table = sweep_goal_table(H, goal_jitter)[goal]
    q0 = np.tile(Q_LINEAR_READY if motion == "linear" else Q_ANGULAR_READY, (count, 1))
    base = platform_state(motion, A, phi)
    ee0 = world_ee_pose(np.asarray(kin.fk(q0)), base)
    p0, o0 = ee0[:, 0:3], ee0[:, 3:6]
    state0 = np.concatenate([q0, np.zeros((count, 6)), p0, o0], axis=1)
    target = np.concatenate([p0, o0], axis=1) + table[:, :6]
    base_traj = np.asarray(kin.predict_base_dynamic(base))
    Jc = np.asarray(kin.jacobian(q0)).reshape(count, 36)
    u_init = dls_guess(Jc, base, p0)
    lim = np.tile(np.asarray(params.v_lim), H)
    x = np.clip(np.tile(u_init, (1, H)) + table[:, 6:], -lim, lim)
    active = ~((np.linalg.norm(base[:, 6:9], axis=1) < 0.01) & (np.linalg.norm(base[:, 9:12], axis=1) < 0.01))
    u_prev = x[:, :6].copy() if with_prev else None
    has_prev = np.ones(count, dtype=np.uint8) if with_prev else None
    hs = HostScenarios(motion, q0, base, state0, target, base_traj, Jc, np.ascontiguousarray(x), u_init,
                       u_prev, has_prev, active)
    hs.amplitude, hs.phase, hs.goal = A, phi, goal
    return hs


class DeviceKinematics:
    """``kin`` backend running on the GPU through libmpc_amd (MpcContext)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def _up(self, a):
        import torch

        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.ctx.device)

    def fk(self, q):
        return self.ctx.fk(self._up(q)).cpu().numpy()

    def jacobian(self, q):
        return self.ctx.jacobian(self._up(q)).cpu().numpy()

    def predict_base_dynamic(self, base):
        return self.ctx.predict_base_dynamic(self._up(base)).cpu().numpy()


def to_device(ctx, hs: HostScenarios):
    """Upload a host batch: returns (ScenarioBatch, x tensor)."""
    import torch

    from .batch import ScenarioBatch

    sc = ScenarioBatch.from_numpy(ctx.device, hs.state0, hs.target, hs.base_traj, hs.Jc, hs.u_prev, hs.has_prev)
    x = torch.as_tensor(hs.x, dtype=torch.float64).to(ctx.device)
    return sc, x
