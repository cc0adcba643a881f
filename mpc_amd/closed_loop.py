"""Headless closed loop: N controllers on N moving platforms, ticking together.

Mirrors the loop of the reference's action server (mpc_action_server.py:53-151) with Gazebo
replaced by the analytic platform motion of tests/simulate_platform_motion.py: the goal pose is
frozen at the pose the arm has when the goal arrives (:77), every tick reads the platform state,
calls ``compute_control`` with zero joint velocities (:80-95), sends ``q + u*dt`` as the next joint
target (:98) and takes the arm to have reached it when the next tick starts (:135).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .batch import MpcContext
from .scenarios import platform_state, world_ee_pose
from .solver import BatchedMPC


@dataclass
class ClosedLoopResult:
    q: np.ndarray          # [ticks+1, N, 6] joint trajectory
    u: np.ndarray          # [ticks, N, 6]   commanded joint velocities
    ee: np.ndarray         # [ticks+1, N, 6] tip pose in the world
    target: np.ndarray     # [N, 6]
    converged: np.ndarray  # [ticks, N] bool


def run_closed_loop(ctx: MpcContext, q0: np.ndarray, motion: str, amplitude: np.ndarray, phase: np.ndarray,
                    ticks: int, solver=None, prep: str = "host") -> ClosedLoopResult:
    """`solver` / `prep` as in :class:`mpc_amd.solver.BatchedMPC` (e.g. a ``SolverPool`` for many controllers)."""
    p = ctx.params
    N = q0.shape[0]
    bm = BatchedMPC(ctx, solver=solver, prep=prep)
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(ctx.device)
    q = np.array(q0, dtype=np.float64)
    base = platform_state(motion, amplitude, phase, 0.0)
    ee = world_ee_pose(ctx.fk(up(q)).cpu().numpy(), base)
    target = ee.copy()                                       # mpc_action_server.py:77
    qs, us, ees, conv = [q.copy()], [], [ee.copy()], []
    for tick in range(ticks):
        base = platform_state(motion, amplitude, phase, tick * p.dt)
        u = bm.compute_control(q, ee, target, base)
        ok = np.ones(N, dtype=bool)
        if bm.last is not None:                              # None: every controller took the early exit this tick
            ok[bm.last_rows] = bm.last.success
        q = q + u * p.dt                                     # mpc_action_server.py:98,135
        base_next = platform_state(motion, amplitude, phase, (tick + 1) * p.dt)
        ee = world_ee_pose(ctx.fk(up(q)).cpu().numpy(), base_next)
        qs.append(q.copy()); us.append(u.copy()); ees.append(ee.copy()); conv.append(ok)
    return ClosedLoopResult(np.stack(qs), np.stack(us), np.stack(ees), target, np.stack(conv))
