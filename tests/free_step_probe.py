"""Run by test_native_slsqp.py::test_free_step_predictions_hold in a process of its own, with MPC_SLSQP_LIB pointing at
the checking build of the host library (libmpc_slsqp_check.so, -DMPC_QP_FREE_STEP=2): whole solves with the team code's
lsq (a team of one) hooked into the C core; the library counts the QPs whose unconstrained step passed lsq's margin test
(the product build returns that step right away) and those among them whose long form did NOT end with an empty final set
after zero NNLS iterations.  Prints one JSON line per case.  usage: free_step_probe.py H N motion [bad rows ...]"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from helpers import host_scenarios, params_for  # noqa: E402
from mpc_amd import native_solver  # noqa: E402
from mpc_amd.native_solver import NativeBatch  # noqa: E402
from mpc_amd.slsqp_state import nz_scatter_index  # noqa: E402
from test_native_slsqp import _rows  # noqa: E402


def main():
    H, N, motion = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    bad = [int(v) for v in sys.argv[4:]]
    lib = native_solver.load()
    mp = params_for(H)
    co, hs = host_scenarios(mp, N, motion=motion, cfg_id=47, goal_jitter=0.0 if H == 20 else 0.05)
    hs.state0[bad, 0] = mp.q_limit + 0.2
    n = mp.n
    lb = np.array([b[0] for b in mp.bounds()])
    ub = np.array([b[1] for b in mp.bounds()])
    scatter = nz_scatter_index(H)
    count = {k: ctypes.c_int.in_dll(lib, "mpc_lsq_spmd_" + k) for k in ("plain_calls", "free_steps", "free_step_errors")}
    lib.mpc_slsqp_set_lsq(ctypes.cast(lib.mpc_lsq_spmd_serial, ctypes.c_void_p))
    ticks = int(os.environ.get("PROBE_TICKS", "1"))      # > 1: further ticks from the shifted solution (the controller's warm start)
    first = int(os.environ.get("PROBE_FIRST", "0"))
    if first:
        co, hs = host_scenarios(mp, N, motion=motion, cfg_id=47, goal_jitter=0.0 if H == 20 else 0.05, first=first)
        hs.state0[bad, 0] = mp.q_limit + 0.2
    x0 = np.tile(hs.u_init, (1, H))
    try:
        for tick in range(ticks):
            b = NativeBatch(H, N, threads=1)
            b.begin(x0, lb, ub, mp.ftol, mp.maxiter)
            f, g, c, a = co.eval_fd(_rows(hs, np.arange(N), b.X[:N]))
            b.F[:N], b.Cc[:N], b.G[:N, :n], b.NZ[:N] = f, c, g, a.reshape(N, -1)[:, scatter]
            while True:
                wf, wg = (w.copy() for w in b.advance())
                if len(wf) + len(wg) == 0:
                    break
                if len(wf):
                    f, c = co.eval(_rows(hs, wf, np.clip(b.X[wf], lb, ub)))
                    b.F[wf], b.Cc[wf] = f, c
                if len(wg):
                    _, g, _, a = co.eval_fd(_rows(hs, wg, np.clip(b.X[wg], lb, ub)))
                    b.G[wg, :n] = g
                    b.NZ[wg] = a.reshape(len(wg), -1)[:, scatter]
            sol = b.X[:N].copy()
            b.close()
            # the next tick as the controller starts it (ur_mpc.py:178-181, 205-212): the solution shifted by one step,
            # the previous control in the smoothness term
            x0 = np.roll(sol, -6, axis=1)
            x0[:, -6:] = hs.u_init
            hs.u_prev = np.ascontiguousarray(sol[:, :6])
            hs.has_prev = np.ones(N, dtype=np.uint8)
    finally:
        lib.mpc_slsqp_set_lsq(None)
    print(json.dumps({"H": H, "N": N, "motion": motion, **{k: int(v.value) for k, v in count.items()}}))


if __name__ == "__main__":
    main()
