"""Process pool for the host-bound half of a batched solve.

Evaluations run on the GPU, but every SLSQP quadratic subproblem runs in scipy's compiled core on
the host, and that core holds the GIL.  ``SolverPool`` therefore spreads the cores of a batch over
worker PROCESSES, while this process stays the only one that talks to the GPU:

* workers own contiguous ranges of scenarios, advance their cores and post "these want (f, c),
  these want (grad f, dc/dx)" in shared memory;
* the server takes whatever requests are ready, gathers those scenarios on the device
  (``mpc_gather_scenarios``), runs ONE ``mpc_eval`` and ONE ``mpc_eval_fd`` launch for all of them,
  and copies the results device -> shared memory, where each worker finds its rows.

Workers are not in lock-step with each other, so device work for one overlaps core work of the
others.  Every problem still sees exactly the numbers the one-process driver would give it, so the
result equals ``BatchedSLSQP.solve`` row for row.  (More than ~8 processes sharing one GPU queue
badly on this stack - measured - which is why the workers do not own contexts of their own.)
"""
from __future__ import annotations

import multiprocessing as mp
import time
from multiprocessing import connection, shared_memory
from typing import List

import numpy as np
import torch

from .batch import EvalResult, MpcContext, ScenarioBatch
from .sharding import shard_range
from .slsqp_state import SharedViews, require_core, worker_main
from .solver import BatchSolveResult


def host_core_budget(local_world: int = 1) -> int:
    """Cores one rank may use.  The CPU affinity and the cgroup CPU quota bound the whole job and
    are shared by the `local_world` ranks of this host; OMP_NUM_THREADS is the launcher's per-process
    hint.  (os.cpu_count() sees the whole machine in a container and is not used.)"""
    import os

    job = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            job = min(job, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    budget = job // max(1, local_world)
    try:
        budget = min(budget, int(os.environ["OMP_NUM_THREADS"]))
    except (KeyError, ValueError):
        pass
    return max(1, budget)


def _host_register(ptr: int, nbytes: int) -> bool:
    import ctypes

    try:
        rc = int(torch.cuda.cudart().cudaHostRegister(ptr, nbytes, 0))
    except Exception:  # noqa: BLE001
        rc = -1
    if rc != 0:
        try:    # a refused registration leaves HIP's sticky last-error set: read it so torch does not trip on it
            ctypes.CDLL("libamdhip64.so").hipGetLastError()
        except OSError:
            pass
    return rc == 0


class SolverPool:
    def __init__(self, ctx: MpcContext, workers: int, capacity: int = None):
        """`capacity`: the largest batch ``solve`` will take (default: the context's max_scenarios)."""
        require_core()
        if workers < 1:
            raise ValueError("workers must be >= 1")
        self.ctx, self.workers = ctx, workers
        self.cap = cap = int(capacity or ctx.max_scenarios)
        if cap > ctx.max_scenarios:
            raise ValueError("capacity larger than the context")
        p = ctx.params
        n, m = p.n, p.m
        self.lb = np.array([b[0] for b in p.bounds()])
        self.ub = np.array([b[1] for b in p.bounds()])
        self.structured = ctx.nz_supported()     # dc/dx crosses PCIe in its structured form where the kernel offers it
        jw = ctx.nz_count if self.structured else n * m
        page = 2 << 20
        size = (SharedViews.nbytes(cap, n, m, jw) + page - 1) // page * page
        self._shm = shared_memory.SharedMemory(create=True, size=size)
        self._v = SharedViews(self._shm.buf, cap, n, m, jw)
        # page-lock the block so that device -> shared-memory copies are direct DMA; without it the
        # copies still work (the runtime stages them), only slower
        base = np.ndarray((self._shm.size,), dtype=np.uint8, buffer=self._shm.buf)
        self._base_ptr = base.ctypes.data
        torch.cuda.init()
        self._registered = _host_register(self._base_ptr, self._shm.size)
        del base
        self._t = {k: torch.from_numpy(getattr(self._v, k)) for k in SharedViews.FIELDS}
        dev = ctx.device
        z = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        self._x_dev = z(cap, n)
        self._out = EvalResult(z(cap), z(cap, n), z(cap, m), None if self.structured else z(cap, n, m),
                               z(cap, jw) if self.structured else None)
        self._sub = {False: ScenarioBatch.empty(dev, cap, p.horizon, False),
                     True: ScenarioBatch.empty(dev, cap, p.horizon, True)}
        spawn = mp.get_context("spawn")      # workers must not inherit a process that holds the GPU
        self._conns, self._procs = [], []
        for _ in range(workers):
            parent, child = spawn.Pipe()
            proc = spawn.Process(target=worker_main, args=(child, self._shm.name, cap, n, m, self.structured), daemon=True)
            proc.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(proc)
        for c in self._conns:
            if c.recv()[0] != "ready":
                self.close()
                raise RuntimeError("solver worker failed to start")

    # ---- device side -------------------------------------------------------------------------
    def _serve(self, sc: ScenarioBatch, reqs, need_grad: bool):
        """reqs: [(first_w, k_w)] with k_w > 0.  One gather + one launch; per-worker copies out."""
        v, ctx = self._v, self.ctx
        lists = [(v.IDXG if need_grad else v.IDXF)[f:f + k] for f, k in reqs]
        idx = np.concatenate(lists)
        K = len(idx)
        if idx.min() < 0 or idx.max() >= sc.count:
            raise RuntimeError("worker posted a scenario index outside the batch")
        idx_dev = torch.from_numpy(idx).to(ctx.device)
        self._x_dev[:K].copy_(torch.from_numpy(np.clip(v.X[idx], self.lb, self.ub)))   # _clip_x_for_func
        sub = ctx.gather(sc, idx_dev, self._sub[sc.u_prev is not None])
        o = self._out
        jac = None
        if need_grad and self.structured:
            ctx.eval_fd_nz(sub, self._x_dev[:K], want_con=False,
                           out=EvalResult(o.cost[:K], o.grad[:K], None, None, o.con_jac_nz[:K]))
            jac = o.con_jac_nz
        elif need_grad:
            ctx.eval_fd(sub, self._x_dev[:K], want_con=False,
                        out=EvalResult(o.cost[:K], o.grad[:K], None, o.con_jac[:K]))
            jac = o.con_jac.view(self.cap, -1)
        else:
            ctx.eval(sub, self._x_dev[:K], out=EvalResult(o.cost[:K], None, o.con[:K], None))
        off = 0
        for f, k in reqs:
            if need_grad:
                self._t["SG"][f:f + k].copy_(o.grad[off:off + k])
                self._t["SJ"][f:f + k].copy_(jac[off:off + k])
            else:
                self._t["SF"][f:f + k].copy_(o.cost[off:off + k])
                self._t["SC"][f:f + k].copy_(o.con[off:off + k])
            off += k
        return K

    def solve(self, sc: ScenarioBatch, x0: np.ndarray, progress=None, progress_every: float = 30.0) -> BatchSolveResult:
        """`progress(seconds, n_eval, n_eval_fd, workers_done)` is called about every `progress_every` seconds."""
        ctx, v = self.ctx, self._v
        p = ctx.params
        N, n = sc.count, p.n
        if not self._procs:
            raise RuntimeError("the solver pool is closed")
        if N > self.cap:
            raise ValueError(f"{N} scenarios > pool capacity {self.cap}")
        if N == 0:
            return BatchSolveResult(np.zeros((0, n)), np.zeros(0), np.zeros(0, int), np.zeros(0, int), 0, 0, 0)
        v.X[:N] = np.clip(np.asarray(x0, dtype=np.float64).reshape(N, n), self.lb, self.ub)   # _slsqp_py.py:268
        spans = [shard_range(N, self.workers, r) for r in range(self.workers)]
        busy = {}
        for c, (first, cnt) in zip(self._conns, spans):
            if cnt:
                c.send(("solve", first, cnt, p.ftol, p.maxiter, self.lb, self.ub))
                busy[c] = (first, cnt)
        t_serve = 0.0
        t0 = time.perf_counter()
        # mode 0 on entry: f, g, c, a at the start point for everyone (_slsqp_py.py:419-425)
        self._x_dev[:N].copy_(torch.from_numpy(v.X[:N]))
        o = self._out
        if self.structured:
            ctx.eval_fd_nz(sc, self._x_dev[:N], out=EvalResult(o.cost[:N], o.grad[:N], o.con[:N], None, o.con_jac_nz[:N]))
            jac = o.con_jac_nz
        else:
            ctx.eval_fd(sc, self._x_dev[:N], out=EvalResult(o.cost[:N], o.grad[:N], o.con[:N], o.con_jac[:N]))
            jac = o.con_jac.view(self.cap, -1)
        for key, src in (("SF", o.cost), ("SC", o.con), ("SG", o.grad), ("SJ", jac)):
            self._t[key][:N].copy_(src[:N])
        n_eval, n_eval_fd = 0, N
        t_serve += time.perf_counter() - t0
        for c in busy:
            c.send(("data",))
        fun, nit, status = np.zeros(N), np.zeros(N, int), np.zeros(N, int)
        t_core = 0.0
        rounds = 0
        t_start = t_last = time.perf_counter()
        while busy:
            ready = connection.wait(list(busy))
            if progress is not None and time.perf_counter() - t_last >= progress_every:
                t_last = time.perf_counter()
                progress(t_last - t_start, n_eval, n_eval_fd, self.workers - len(busy))
            t0 = time.perf_counter()
            req_f, req_g, answered = [], [], []
            for c in ready:
                try:
                    msg = c.recv()
                except (EOFError, OSError) as exc:       # a worker died (killed, out of memory)
                    self._abort(busy)
                    raise RuntimeError("a solver worker process ended unexpectedly; the pool is closed") from exc
                first, cnt = busy[c]
                if msg[0] == "done":
                    fun[first:first + cnt], nit[first:first + cnt], status[first:first + cnt] = msg[1], msg[2], msg[3]
                    t_core = max(t_core, msg[4])
                    rounds = max(rounds, msg[5])
                    del busy[c]
                elif msg[0] == "req":
                    if msg[1]:
                        req_f.append((first, msg[1]))
                    if msg[2]:
                        req_g.append((first, msg[2]))
                    answered.append(c)
                else:
                    self._abort(busy)
                    raise RuntimeError(f"solver worker failed: {msg[1] if len(msg) > 1 else msg}")
            if req_f:
                n_eval += self._serve(sc, req_f, False)
            if req_g:
                n_eval_fd += self._serve(sc, req_g, True)
            for c in answered:
                c.send(("data",))
            t_serve += time.perf_counter() - t0
        return BatchSolveResult(x=v.X[:N].copy(), fun=fun, nit=nit, status=status, rounds=rounds, n_eval=n_eval,
                                n_eval_fd=n_eval_fd, t_core=t_core, t_serve=t_serve)

    def _abort(self, busy):
        """Give up on the running solve: the workers hold half-advanced cores, so the pool is closed."""
        busy.clear()
        for proc in self._procs:
            if proc.is_alive():
                proc.terminate()
        self.close()

    def close(self):
        for c in self._conns:
            try:
                c.send(("stop",))
            except Exception:  # noqa: BLE001
                pass
        for proc in self._procs:
            proc.join(timeout=10)
            if proc.is_alive():
                proc.terminate()
        self._conns, self._procs = [], []
        if getattr(self, "_shm", None) is not None:
            self._t, self._v = {}, None
            if self._registered:
                try:
                    torch.cuda.cudart().cudaHostUnregister(self._base_ptr)
                except Exception:  # noqa: BLE001
                    pass
            try:
                self._shm.close()
            except BufferError:
                pass        # a view is still alive somewhere; the segment is unlinked regardless
            self._shm.unlink()
            self._shm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
