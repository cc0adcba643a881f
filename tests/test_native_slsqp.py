"""CPU: the native SLSQP core and batched driver (libmpc_slsqp.so) against scipy's compiled core.
The C oracle serves the evaluations (test infrastructure; the product serves them from libmpc_amd)."""
import numpy as np
import pytest

from helpers import host_scenarios, params_for
from mpc_amd.native_solver import NativeBatch
from mpc_amd.slsqp_state import Problem, nz_scatter_index


def _row(hs, i, x):
    d = hs.as_oracle_dict(i, i + 1)
    d["x"] = np.asarray(x, float).reshape(1, -1)
    return d


def _rows(hs, idx, X):
    d = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    d["x"] = np.ascontiguousarray(X)
    return d


def _scipy_core_solve(co, hs, i, mp, lb, ub, x0):
    """scipy's own compiled core, driven like _minimize_slsqp does; returns the per-call history."""
    n, m = mp.n, mp.m
    A = np.zeros((n + 1, m))
    p = Problem(np.clip(x0, lb, ub).copy(), n, m, mp.ftol, mp.maxiter, A.T)
    f, g, c, a = co.eval_fd(_row(hs, i, p.x))
    p.fx, p.c[:], p.g[:n], A[:n] = float(f[0]), c[0], g[0], a[0]
    hist = []
    while not p.done:
        md = p.step(lb, ub)
        hist.append((md, p.x.copy()))
        xc = np.clip(p.x, lb, ub)
        if md == 1:
            f, c = co.eval(_row(hs, i, xc))
            p.fx, p.c[:] = float(f[0]), c[0]
        elif md == -1:
            _, g, _, a = co.eval_fd(_row(hs, i, xc))
            p.g[:n], A[:n] = g[0], a[0]
    return p, hist


@pytest.mark.parametrize("H,N,motion,prev,bad", [(1, 12, "linear", False, ()), (3, 10, "angular", True, ()),
                                                (20, 5, "angular", False, ()), (40, 2, "linear", True, ()),
                                                (3, 6, "angular", False, (0, 3, 4)), (20, 3, "linear", False, (1,))])
def test_native_batch_reproduces_scipy_core_call_by_call(H, N, motion, prev, bad):
    """Every round of the native driver leaves every problem exactly where scipy's core leaves it after the
    same number of calls: same mode, bit-identical x; same exit mode, iteration count and f at the end.
    `bad`: scenarios that start with a joint beyond its limit (inconsistent linearisation: Kraft's augmented problem)."""
    mp = params_for(H)
    co, hs = host_scenarios(mp, N, motion=motion, cfg_id=41, with_prev=prev, goal_jitter=0.0 if H == 20 else 0.05)
    hs.state0[list(bad), 0] = mp.q_limit + 0.2
    n, m = mp.n, mp.m
    lb = np.array([b[0] for b in mp.bounds()])
    ub = np.array([b[1] for b in mp.bounds()])
    x0 = np.tile(hs.u_init, (1, H))
    want = [_scipy_core_solve(co, hs, i, mp, lb, ub, x0[i]) for i in range(N)]
    scatter = nz_scatter_index(H)
    b = NativeBatch(H, N + 3, threads=3)
    b.begin(x0, lb, ub, mp.ftol, mp.maxiter)
    f, g, c, a = co.eval_fd(_rows(hs, np.arange(N), b.X[:N]))
    b.F[:N], b.Cc[:N], b.G[:N, :n], b.NZ[:N] = f, c, g, a.reshape(N, -1)[:, scatter]
    calls = np.zeros(N, int)
    running = np.ones(N, bool)
    while running.any():
        want_f, want_g = (w.copy() for w in b.advance())
        for i in np.nonzero(running)[0]:
            md, x = want[i][1][calls[i]]
            assert int(b.mode[i]) == md
            np.testing.assert_array_equal(b.X[i], x)
            calls[i] += 1
            if md not in (1, -1):
                running[i] = False
        assert sorted(np.concatenate([want_f, want_g])) == list(np.nonzero(running)[0])
        if len(want_f):
            f, c = co.eval(_rows(hs, want_f, np.clip(b.X[want_f], lb, ub)))
            b.F[want_f], b.Cc[want_f] = f, c
        if len(want_g):
            _, g, _, a = co.eval_fd(_rows(hs, want_g, np.clip(b.X[want_g], lb, ub)))
            b.G[want_g, :n] = g
            b.NZ[want_g] = a.reshape(len(want_g), -1)[:, scatter]
    for i in range(N):
        p = want[i][0]
        assert int(b.mode[i]) == int(p.mode) and int(b.iterations[i]) == int(p.majiter)
        assert b.F[i] == p.fx
        assert calls[i] == len(want[i][1])
    assert b.core_seconds > 0.0
    b.close()


def test_native_batch_argument_checks():
    with pytest.raises(ValueError):
        NativeBatch(0, 4)
    b = NativeBatch(1, 2, threads=1)
    with pytest.raises(ValueError):
        b.begin(np.zeros((3, 6)), -np.ones(6), np.ones(6), 1e-3, 50)      # more problems than capacity
    b.begin(np.full((2, 6), 5.0), -np.ones(6), np.ones(6), 1e-3, 50)
    np.testing.assert_array_equal(b.X, np.ones((2, 6)))                   # start point clipped to the bounds
    b.close()


@pytest.mark.parametrize("H,N,bad", [(1, 8, ()), (3, 8, ()), (20, 4, ()), (1, 6, (1, 4)), (3, 8, (2, 5)), (20, 4, (1,))])
def test_team_parallel_lsq_with_a_team_of_one_keeps_whole_solves_identical(H, N, bad):
    """lsq_spmd.hpp is the QP subproblem written for a team of threads (the form a GPU workgroup runs).  With a
    team of one, hooked into the C core in place of its own LSQ, whole solves must not move by a bit.
    `bad`: scenarios whose first joint starts beyond its limit: a constraint row is negative with a zero gradient, the
    linearisation is inconsistent and the body turns to Kraft's augmented problem (one more variable), which the team
    code solves as well (lsq<AUG>)."""
    import ctypes

    from mpc_amd import native_solver

    lib = native_solver.load()
    mp = params_for(H)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=42, goal_jitter=0.0 if H == 20 else 0.05)
    hs.state0[list(bad), 0] = mp.q_limit + 0.2
    aug_calls = ctypes.c_int.in_dll(lib, "mpc_lsq_spmd_augmented_calls")
    aug_before = aug_calls.value
    diag_calls = ctypes.c_int.in_dll(lib, "mpc_lsq_spmd_diagonal_calls")       # lsq's short forms for a diagonal factor
    diag_before = diag_calls.value
    n = mp.n
    lb = np.array([b[0] for b in mp.bounds()])
    ub = np.array([b[1] for b in mp.bounds()])
    scatter = nz_scatter_index(H)

    def solve():
        b = NativeBatch(H, N, threads=1)
        b.begin(np.tile(hs.u_init, (1, H)), lb, ub, mp.ftol, mp.maxiter)
        f, g, c, a = co.eval_fd(_rows(hs, np.arange(N), b.X[:N]))
        b.F[:N], b.Cc[:N], b.G[:N, :n], b.NZ[:N] = f, c, g, a.reshape(N, -1)[:, scatter]
        trace = []
        while True:
            wf, wg = (w.copy() for w in b.advance())
            trace.append((b.X[:N].copy(), b.mode[:N].copy()))
            if len(wf) + len(wg) == 0:
                break
            if len(wf):
                f, c = co.eval(_rows(hs, wf, np.clip(b.X[wf], lb, ub)))
                b.F[wf], b.Cc[wf] = f, c
            if len(wg):
                _, g, _, a = co.eval_fd(_rows(hs, wg, np.clip(b.X[wg], lb, ub)))
                b.G[wg, :n] = g
                b.NZ[wg] = a.reshape(len(wg), -1)[:, scatter]
        out = (trace, b.iterations[:N].copy(), b.F[:N].copy())
        b.close()
        return out

    want = solve()
    lib.mpc_slsqp_set_lsq(ctypes.cast(lib.mpc_lsq_spmd_serial, ctypes.c_void_p))
    lib.mpc_slsqp_set_bfgs(ctypes.cast(lib.mpc_bfgs_spmd_serial, ctypes.c_void_p))      # the BFGS / LDL' update likewise
    try:
        got = solve()
    finally:
        lib.mpc_slsqp_set_lsq(None)
        lib.mpc_slsqp_set_bfgs(None)
    assert (aug_calls.value > aug_before) == bool(bad)
    assert diag_calls.value - diag_before >= N                 # every solve's first QP at least
    assert len(got[0]) == len(want[0])
    for (xa, ma), (xb, mb) in zip(got[0], want[0]):
        np.testing.assert_array_equal(ma, mb)
        np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[2], want[2])


@pytest.mark.parametrize("H,N,motion,bad,ticks", [(20, 5, "angular", (), 1), (3, 60, "angular", (2, 5), 1), (3, 60, "linear", (), 3),
                                                  (1, 80, "angular", (0, 3), 2), (20, 3, "linear", (), 2)])
def test_free_step_predictions_hold(H, N, motion, bad, ticks):
    """lsq's free step (lsq_spmd.hpp, MPC_QP_FREE_STEP): where the unconstrained step x = e^-1 f satisfies every row of
    the constraints with a margin, the product returns it without forming the transformed matrix or entering NNLS,
    because the long form would find no positive h and end with that x and zero multipliers.  The checking build of the
    host library (-DMPC_QP_FREE_STEP=2) runs the long form all the same and counts the predictions that did not hold:
    none may, and most QPs of a solve take the free step (so the test says something).  ticks > 1: further solves from
    the shifted solution with the previous control in the smoothness term, the controller's warm start."""
    import json
    import os
    import subprocess
    import sys

    from mpc_amd import build

    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "free_step_probe.py")
    env = dict(os.environ, MPC_SLSQP_LIB=build.SLSQP_CHECK_LIB_PATH, PROBE_TICKS=str(ticks))
    assert os.path.exists(build.SLSQP_CHECK_LIB_PATH), "python -m mpc_amd.build makes libmpc_slsqp_check.so"
    out = subprocess.run([sys.executable, probe, str(H), str(N), motion] + [str(v) for v in bad], env=env, check=True,
                         capture_output=True, text=True, timeout=600).stdout
    got = json.loads(out.strip().splitlines()[-1])
    assert got["free_step_errors"] == 0
    assert got["plain_calls"] > N and got["free_steps"] > 0.5 * got["plain_calls"]


@pytest.mark.parametrize("H,N,bad", [(3, 8, (2, 5)), (20, 3, (1,)), (1, 6, (0, 3))])
def test_offload_protocol_with_the_augmented_problem_keeps_whole_solves_identical(H, N, bad):
    """The QP-offload form of the body (slsqp_body.inc with S->offload = 2, the form the device-resident solve runs)
    against the whole method on the host, call by call: the body asks for its QPs, a team of one serves them through
    mpc_qp_step_serial (the text of the device's qp_step_kernel: reset / BFGS update of the factor, the QP, the
    multipliers and the Lagrangian gradient kept with the factor), and where a linearisation is inconsistent the body
    asks for Kraft's augmented problem, up to six times with growing weight.  Same modes and iterates in every round
    that both forms have (the offloaded one has the QP rounds in between), same exit."""
    import ctypes as C

    from mpc_amd import native_solver

    lib = native_solver.load()
    mp = params_for(H)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=43, goal_jitter=0.0 if H == 20 else 0.05)
    hs.state0[list(bad), 0] = mp.q_limit + 0.2
    n, m = mp.n, mp.m
    lb = np.array([b[0] for b in mp.bounds()])
    ub = np.array([b[1] for b in mp.bounds()])
    scatter = nz_scatter_index(H)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    lib.mpc_qp_step_serial.restype = C.c_int
    lib.mpc_qp_step_serial.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 9

    def solve(offload):
        b = NativeBatch(H, N, threads=1, offload_qp=offload)
        b.begin(np.tile(hs.u_init, (1, H)), lb, ub, mp.ftol, mp.maxiter)
        f, g, c, a = co.eval_fd(_rows(hs, np.arange(N), b.X[:N]))
        b.F[:N], b.Cc[:N], b.G[:N, :n], b.NZ[:N] = f, c, g, a.reshape(N, -1)[:, scatter]
        L = np.zeros((N, (n + 1) * n // 2 + 1))
        V = np.zeros((N, n))
        trace, flags_seen = [], set()
        while True:
            if offload:
                wf, wg, wq = (w.copy() for w in b.advance_qp())
            else:
                wf, wg = (w.copy() for w in b.advance())
                wq = np.zeros(0, np.int32)
            trace.append((b.X[:N].copy(), b.mode[:N].copy()))
            if len(wf) + len(wg) + len(wq) == 0:
                break
            if len(wf):
                f, c = co.eval(_rows(hs, wf, np.clip(b.X[wf], lb, ub)))
                b.F[wf], b.Cc[wf] = f, c
            if len(wg):
                _, g, _, a = co.eval_fd(_rows(hs, wg, np.clip(b.X[wg], lb, ub)))
                b.G[wg, :n] = g
                b.NZ[wg] = a.reshape(len(wg), -1)[:, scatter]
            for p in wq:
                fl = int(b.QPFLAGS[p])
                flags_seen.add(fl)
                lo, hi, s = (np.ascontiguousarray(v[p, :n]) for v in (b.LO, b.HI, b.S))
                g = np.ascontiguousarray(b.G[p, :n])
                md = lib.mpc_qp_step_serial(H, fl, dp(lo), dp(hi), dp(s), dp(L[p]), dp(g), dp(b.Cc[p]), dp(b.NZ[p]), dp(b.R[p]),
                                            dp(V[p]))
                b.QPMODE[p] = md
                if md == 1:
                    b.S[p, :n] = s
        out = (trace, b.iterations[:N].copy(), b.F[:N].copy(), flags_seen)
        b.close()
        return out

    want = solve(False)
    got = solve(2)
    assert any(fl & 4 for fl in got[3]), "no augmented problem was met"
    # the offloaded form's rounds that ask for a QP are extra; in every other round both forms stand at the same point
    QP = 20
    mine = [(x, md) for x, md in got[0]]
    k = np.zeros(N, int)                                   # per problem: position in the host form's trace
    host_modes = np.stack([md for _, md in want[0]])
    host_x = np.stack([x for x, _ in want[0]])
    done = np.zeros(N, bool)
    for x, md in mine:
        for i in range(N):
            if done[i] or md[i] == QP:
                continue
            assert md[i] == host_modes[k[i], i]
            np.testing.assert_array_equal(x[i], host_x[k[i], i])
            if md[i] not in (1, -1):
                done[i] = True
            else:
                k[i] += 1
    assert done.all()
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[2], want[2])


def test_offload_staging_packs_and_unpacks_rows():
    """mpc_slsqp_pack / mpc_slsqp_unpack (QP offload): one round's requests gathered into / scattered from flat
    staging buffers with the layouts include/mpc_slsqp.h gives."""
    import ctypes as C

    H, N = 3, 7
    n, m = 6 * H, 12 * (H + 1)
    rng = np.random.default_rng(3)
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    b = NativeBatch(H, N, threads=2, offload_qp=True)
    b.begin(rng.uniform(-0.4, 0.4, (N, n)), lb, ub, 1e-3, 50)
    b.X[:N] += rng.uniform(-0.3, 0.3, (N, n))             # some entries outside the box: pack must clip them
    b.LO[:N, :n], b.HI[:N, :n], b.S[:N, :n] = rng.normal(size=(3, N, n))
    flags = np.ctypeslib.as_array(b._lib.mpc_slsqp_qp_flags(b._b), shape=(N,))
    wf, wg, wq = np.array([5, 0], np.int32), np.array([2], np.int32), np.array([6, 1, 3], np.int32)
    nf, ng, nq = len(wf), len(wg), len(wq)
    stage, istage = np.full((nf + ng + 3 * nq) * n, np.nan), np.full(nf + ng + 2 * nq, -7, np.int32)
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    assert b._lib.mpc_slsqp_pack(b._b, nf, i32(wf), ng, i32(wg), nq, i32(wq), stage.ctypes.data, istage.ctypes.data) == 0
    want = np.concatenate([np.clip(b.X[wf], lb, ub).ravel(), np.clip(b.X[wg], lb, ub).ravel(), b.LO[wq, :n].ravel(),
                           b.HI[wq, :n].ravel(), b.S[wq, :n].ravel()])
    np.testing.assert_array_equal(stage, want)
    np.testing.assert_array_equal(istage, np.concatenate([wf, wg, wq, flags[wq]]))
    # the way back: c, grad f, step, multipliers, f; a step is taken only where the QP's mode is 1
    back = rng.normal(size=nf * m + ng * n + nq * (n + m) + nf)
    modes = np.array([1, 4, 1], np.int32)
    s_before, r_before = b.S[:N].copy(), b.R[:N].copy()
    assert b._lib.mpc_slsqp_unpack(b._b, nf, i32(wf), ng, i32(wg), nq, i32(wq), back.ctypes.data, modes.ctypes.data) == 0
    c, g, s, r, f = np.split(back, np.cumsum([nf * m, ng * n, nq * n, nq * m]))
    np.testing.assert_array_equal(b.Cc[wf], c.reshape(nf, m))
    np.testing.assert_array_equal(b.F[wf], f)
    np.testing.assert_array_equal(b.G[wg, :n], g.reshape(ng, n))
    np.testing.assert_array_equal(b.QPMODE[wq], modes)
    ok = modes == 1
    np.testing.assert_array_equal(b.S[wq[ok], :n], s.reshape(nq, n)[ok])
    np.testing.assert_array_equal(b.R[wq[ok], :m], r.reshape(nq, m)[ok])
    np.testing.assert_array_equal(b.S[wq[~ok]], s_before[wq[~ok]])
    np.testing.assert_array_equal(b.R[wq[~ok]], r_before[wq[~ok]])
    assert b._lib.mpc_slsqp_pack(b._b, -1, i32(wf), 0, i32(wg), 0, i32(wq), stage.ctypes.data, istage.ctypes.data) == -1
    b.close()
    plain = NativeBatch(H, N, threads=1)                   # not in offload mode: refused
    assert plain._lib.mpc_slsqp_pack(plain._b, 0, i32(wf), 0, i32(wg), 0, i32(wq), stage.ctypes.data, istage.ctypes.data) == -1
    plain.close()
