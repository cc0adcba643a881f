"""Batched SLSQP with every problem resident on the GPU (``mpc_solve_*`` in include/mpc_amd.h).

scipy's driver loop (``_slsqp_py.py:427-456``, called by ``URMPC.compute_control`` at ur_mpc.py:188-200) runs
on the device from the start point to the solution: Kraft's O(n + m) body as a one-thread-per-problem kernel
(the text the host core of ``libmpc_slsqp.so`` is compiled from), the evaluations and the BFGS update + QP of
each round driven by device-side request lists.  This process only enqueues rounds and, every few rounds,
reads ONE integer per lane (how many problems are unfinished) from pinned memory: no host threads run SLSQP
logic, nothing is staged across PCIe per iteration.  Results equal the host drivers' bit for bit
(tests/test_gpu_solve.py::test_resident_driver_equals_host_driver).

A batch can be split over up to ``lanes`` independent parts, each with its own stream, context (QP workspace) and
state, enqueued round-robin from this one thread.  Measured on MI355X one lane is fastest at every batch size
(65,536 problems: 1.43 s on one lane, 1.64-1.72 s on two to four): the QP teams take their requests from a
device-side queue, so a single launch per round already keeps every CU busy until the round's last QP, and parts
only compete for the CUs' LDS.  More lanes remain available for callers that want to interleave other work.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _ffi
from .batch import MpcContext, ScenarioBatch
from .native_solver import MODE_NEEDS_HOST, active_lanes
from .solver import BatchSolveResult


def p_supports_team(ctx: MpcContext) -> bool:
    """mpc_solve_run's conditions (include/mpc_amd.h): the structured dc/dx path and a horizon of at most 21."""
    return ctx.nz_supported() and ctx.params.horizon <= 21


class _ResidentLane:
    def __init__(self, ctx: MpcContext, capacity: int):
        self.ctx = ctx
        self.cap = cap = int(capacity)
        p = ctx.params
        dev = ctx.device
        self.ws_bytes = int(ctx._lib.mpc_solve_workspace_bytes(p.horizon, cap))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.x0 = torch.empty(cap, p.n, dtype=torch.float64, device=dev)
        self.x = torch.empty(cap, p.n, dtype=torch.float64, device=dev)
        self.fun = torch.empty(cap, dtype=torch.float64, device=dev)
        self.nit = torch.empty(cap, dtype=torch.int32, device=dev)
        self.status = torch.empty(cap, dtype=torch.int32, device=dev)
        self.stats = torch.zeros(5, dtype=torch.int32, device=dev)
        self.left = torch.zeros(2, dtype=torch.int32).pin_memory()      # "unfinished" of the two batches in flight
        self.events = [torch.cuda.Event(), torch.cuda.Event()]
        self.stream = torch.cuda.Stream(device=dev)

    def _call(self, fn, *args):
        _ffi.check(self.ctx._ctx, fn(self.ctx._ctx, *args))

    def begin(self, sc: ScenarioBatch, x0):
        """x0: [N, n] start points, a numpy array or a tensor already on the device."""
        ctx, p = self.ctx, self.ctx.params
        self.sc, self.N = sc, sc.count
        self.scc = ctx._scen_c(sc)
        self.bound, self.batches, self.rounds, self.running = self.N, 0, 0, True
        with torch.cuda.stream(self.stream):
            src = x0 if isinstance(x0, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x0))
            self.x0[:self.N].copy_(src, non_blocking=True)
            self._call(ctx._lib.mpc_solve_begin, C.byref(self.scc), self.x0.data_ptr(), float(p.ftol), int(p.maxiter),
                       self.ws.data_ptr(), self.cap, self.ws_bytes, C.c_void_p(self.stream.cuda_stream))

    def enqueue(self, rounds: int):
        """`rounds` more rounds on the lane's stream; the count of problems unfinished after them lands in pinned
        memory, followed by an event."""
        slot = self.batches & 1
        self._call(self.ctx._lib.mpc_solve_rounds, C.byref(self.scc), self.ws.data_ptr(), self.cap, self.ws_bytes, int(rounds),
                   int(self.bound), self.left[slot:].data_ptr(), C.c_void_p(self.stream.cuda_stream))
        self.events[slot].record(self.stream)
        self.batches += 1
        self.rounds += rounds

    def run_to_the_end(self):
        """Every problem to its end in one launch (mpc_solve_run: a workgroup per problem at a time, no lock step)."""
        p = self.ctx.params
        self._call(self.ctx._lib.mpc_solve_run, C.byref(self.scc), self.ws.data_ptr(), self.cap, self.ws_bytes, int(p.maxiter),
                   self.left.data_ptr(), C.c_void_p(self.stream.cuda_stream))
        self.events[0].record(self.stream)
        self.events[0].synchronize()
        if int(self.left[0]) != 0:
            raise RuntimeError(f"mpc_solve_run gave up on {int(self.left[0])} problems (bound on calls of the body)")
        self.running = False

    def check(self):
        """Wait for the batch before the one just enqueued; its count bounds every later round's lists."""
        if self.batches < 2:
            return
        slot = self.batches & 1                   # the older of the two in flight
        self.events[slot].synchronize()
        self.bound = int(self.left[slot])
        if self.bound < 0:
            raise RuntimeError("mpc_solve_rounds met more requests than its `bound` sized the grids for: the solve is void")
        if self.bound == 0:
            self.running = False

    def finish(self, x_on_device: bool = False):
        """x_on_device: the solutions stay where they are (a view of the lane's buffer, valid until its next solve)."""
        with torch.cuda.stream(self.stream):
            self._call(self.ctx._lib.mpc_solve_result, self.N, self.ws.data_ptr(), self.cap, self.ws_bytes, self.x.data_ptr(),
                       self.fun.data_ptr(), self.nit.data_ptr(), self.status.data_ptr(), self.stats.data_ptr(),
                       C.c_void_p(self.stream.cuda_stream))
            N = self.N
            out = (self.x[:N] if x_on_device else self.x[:N].cpu().numpy(), self.fun[:N].cpu().numpy(),
                   self.nit[:N].cpu().numpy().astype(int), self.status[:N].cpu().numpy().astype(int),
                   self.stats.cpu().numpy().astype(int))
        self.sc = self.scc = None
        return out


class ResidentSLSQP:
    """Same interface as the other batched drivers (``solve(ScenarioBatch, x0) -> BatchSolveResult``)."""

    def __init__(self, ctx: MpcContext, capacity: int = None, lanes: int = 1, min_per_lane: int = 2048,
                 rounds_per_check: int = 6, team_max: int = None):
        """team_max: batches of up to that many problems are solved without the lock step (mpc_solve_run: one launch,
        a workgroup per problem at a time; same results bit for bit), larger ones in lock-step rounds.  None: 12,288
        where the library supports it for this horizon (H <= 21), 0: never."""
        if not ctx.nz_supported():
            raise ValueError("the driver takes dc/dx in its structured form: the context needs MPC_ALGO_PREFIX")
        if not 1 <= lanes <= 8:
            raise ValueError("lanes must be 1..8")
        self.ctx = ctx
        self.cap = cap = int(capacity or ctx.max_scenarios)
        self.min_per_lane = max(1, int(min_per_lane))
        self.rounds_per_check = max(1, int(rounds_per_check))
        lanes = max(1, min(lanes, cap // self.min_per_lane))
        lane_cap = (cap + lanes - 1) // lanes
        self._ctxs = [ctx] + [MpcContext(ctx.params, lane_cap, ctx.device_index) for _ in range(lanes - 1)]
        self.lanes = [_ResidentLane(c, lane_cap) for c in self._ctxs]
        self.threads = 1
        if team_max is None:
            team_max = 12288 if p_supports_team(ctx) else 0
        self.team_max = int(team_max)

    device_io = True       # solve() takes start points that are already on the device and can leave the solutions there

    def solve(self, sc: ScenarioBatch, x0, x_on_device: bool = False) -> BatchSolveResult:
        """x0 [N, n]: numpy, or a float64 tensor on the context's device (no copy through the host).
        x_on_device: the result's ``x`` is None and ``x_dev`` is a device tensor [N, n] (valid until the next solve):
        a caller that needs the first control of every solution and a warm start for the next tick moves 48 of the
        960 bytes per controller."""
        ctx = self.ctx
        p = ctx.params
        N, n = sc.count, p.n
        if N > self.cap:
            raise ValueError(f"{N} scenarios > capacity {self.cap}")
        if N == 0:
            res = BatchSolveResult(np.zeros((0, n)), np.zeros(0), np.zeros(0, int), np.zeros(0, int), 0, 0, 0)
            res.x_dev = torch.zeros(0, n, dtype=torch.float64, device=ctx.device) if x_on_device else None
            return res
        if isinstance(x0, torch.Tensor):
            if x0.dtype != torch.float64 or x0.device != ctx.device or tuple(x0.shape) != (N, n) or not x0.is_contiguous():
                raise ValueError("x0 on the device must be a contiguous float64 [N, n] tensor on the context's device")
        else:
            x0 = np.asarray(x0, dtype=np.float64).reshape(N, n)
        t_start = time.perf_counter()
        caller = torch.cuda.current_stream(ctx.device)
        k = active_lanes(N, len(self.lanes), self.lanes[0].cap, self.min_per_lane)
        cuts = [(N * i) // k for i in range(k + 1)]
        lanes = self.lanes[:k]
        for i, lane in enumerate(lanes):
            lane.stream.wait_stream(caller)              # the scenarios were written on the caller's stream
            lane.begin(sc.slice(cuts[i], cuts[i + 1]), x0[cuts[i]:cuts[i + 1]])
        R = self.rounds_per_check
        if N <= self.team_max:
            for lane in lanes:
                with torch.cuda.stream(lane.stream):
                    lane.run_to_the_end()
        while any(lane.running for lane in lanes):
            for lane in lanes:
                if lane.running:
                    lane.enqueue(R)
            for lane in lanes:
                if lane.running:
                    lane.check()
        parts = [lane.finish(x_on_device) for lane in lanes]
        for lane in lanes:
            caller.wait_stream(lane.stream)
        if x_on_device:
            x_dev = parts[0][0] if len(parts) == 1 else torch.cat([r[0] for r in parts])
            x = None
        else:
            x_dev = None
            x = np.concatenate([r[0] for r in parts])
        fun = np.concatenate([r[1] for r in parts])
        nit = np.concatenate([r[2] for r in parts])
        status = np.concatenate([r[3] for r in parts])
        stats = np.sum([r[4] for r in parts], axis=0)
        n_eval, n_eval_fd, n_qp = int(stats[0]), int(stats[1]) + N, int(stats[2])
        if (status == MODE_NEEDS_HOST).any():      # the device solves Kraft's augmented problem itself: never handed back
            raise RuntimeError("status 104 from a device-resident solve: the library and this driver are out of step")
        res = BatchSolveResult(x=x, fun=fun, nit=nit, status=status, rounds=max(lane.rounds for lane in lanes), n_eval=n_eval,
                               n_eval_fd=n_eval_fd, t_core=0.0, t_serve=time.perf_counter() - t_start)
        res.n_qp, res.t_qp, res.n_host_fallback = n_qp, 0.0, 0
        res.n_free_steps = int(stats[4])       # QPs whose unconstrained step sufficed (lsq's free step)
        res.x_dev = x_dev
        res.t_wall = time.perf_counter() - t_start
        return res

    def close(self):
        self.lanes = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
