"""GPU: BASELINE configs[2]'s solve loop (H=20, full SLSQP solves) pinned to the REAL reference.

tests/golden/solve_H20_cfg3.npz holds two compute_control ticks of the reference for the first active rows
of the very scenario list bench.py's solves leg runs (angular, cfg_id 3, goal_jitter 0) plus linear ones;
tests/golden/closed_loop_H20.npz a 4-tick closed-loop trajectory per motion type (oracle/gen_golden.py --h20).
Every driver of this repo — the drop-in class, the lock-step driver over scipy's core, the native host
driver and DeviceQpSLSQP (the one bench.py times) — is compared with them DIRECTLY at the north star's
1e-6 rad, and the full-size batch goes through size-independent properties."""
import os

import numpy as np
import pytest
import torch

from mpc_amd.batch import MpcContext
from mpc_amd.params import MpcParams

pytestmark = pytest.mark.gpu

SOL_ATOL = 1e-6   # north star: converged joint trajectory within 1e-6 rad


def _golden(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.fail(f"{name} missing: python -m oracle.gen_golden --h20 (where /root/reference exists)")
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    return np.load(path, allow_pickle=False)


def _solver(kind, ctx, N):
    from mpc_amd.native_solver import DeviceQpSLSQP, NativeBatchedSLSQP

    if kind == "scipy_core":
        return None
    if kind == "native":
        return NativeBatchedSLSQP(ctx, threads=3, capacity=N)
    if kind == "resident":
        from mpc_amd.device_solver import ResidentSLSQP

        return ResidentSLSQP(ctx, capacity=N, lanes=2, min_per_lane=2)
    return DeviceQpSLSQP(ctx, threads=3, capacity=N, lanes=2, min_per_lane=2)     # two lanes even for 7 problems


@pytest.mark.parametrize("kind", ["scipy_core", "native", "device_qp", "resident"])
def test_h20_solves_match_reference(golden_dir, kind):
    from mpc_amd.solver import BatchedMPC

    g = _golden(golden_dir, "solve_H20_cfg3.npz")
    N = g["q0"].shape[0]
    assert N >= 6 and (g["motion"] == "angular").sum() >= 3 and (g["motion"] == "linear").sum() >= 3
    ctx = MpcContext(MpcParams.reference_yaml_defaults(horizon=20), max_scenarios=N, device=0)
    solver = _solver(kind, ctx, N)
    try:
        bm = BatchedMPC(ctx, solver=solver)
        for tick in range(g["control"].shape[1]):          # tick 1: warm start, damping term, EMA
            u = bm.compute_control(g["q0"], g["ee"], g["target"], g["base"])
            assert bm.last.success.all()
            np.testing.assert_allclose(bm.previous_solution, g["solution"][:, tick], rtol=0, atol=SOL_ATOL)
            np.testing.assert_allclose(u, g["control"][:, tick], rtol=0, atol=SOL_ATOL)
    finally:
        if solver is not None:
            solver.close()


@pytest.mark.parametrize("solver", ["scipy", "resident"])
def test_h20_drop_in_class_matches_reference(golden_dir, solver):
    """GpuURMPC.compute_control against the reference's own two-tick solves (ur_mpc.py:83-247), with scipy's
    minimize around the device callbacks and with minimize replaced by the device-resident solver."""
    from mpc_amd.ur_mpc import GpuURMPC

    g = _golden(golden_dir, "solve_H20_cfg3.npz")
    rows = (0, g["q0"].shape[0] - 1) if solver == "scipy" else range(g["q0"].shape[0])   # scipy drives these: slow
    for i in rows:
        mpc = GpuURMPC(MpcParams.reference_yaml_defaults(horizon=20), solver=solver)
        b = g["base"][i]
        args = ({"position": g["q0"][i].copy(), "velocity": np.zeros(6)},
                {"position": g["ee"][i, :3].copy(), "orientation": g["ee"][i, 3:].copy()},
                {"position": g["target"][i, :3].copy(), "orientation": g["target"][i, 3:].copy()},
                {"position": b[0:3].copy(), "orientation": b[3:6].copy(), "linear_velocity": b[6:9].copy(),
                 "angular_velocity": b[9:12].copy()})
        for tick in range(2):
            u = mpc.compute_control(*args)
            np.testing.assert_allclose(mpc.previous_solution, g["solution"][i, tick], rtol=0, atol=SOL_ATOL)
            np.testing.assert_allclose(u, g["control"][i, tick], rtol=0, atol=SOL_ATOL)


@pytest.mark.parametrize("kind", ["scipy_core", "device_qp", "resident"])
def test_h20_closed_loop_matches_reference(golden_dir, kind):
    """Four ticks of the action server's loop at H=20 against the trajectory the real reference produced."""
    from mpc_amd.closed_loop import run_closed_loop

    g = _golden(golden_dir, "closed_loop_H20.npz")
    ctx = MpcContext(MpcParams.reference_yaml_defaults(horizon=20), max_scenarios=4, device=0)
    for i, motion in enumerate(g["motion"]):
        solver = _solver(kind, ctx, 4)
        try:
            got = run_closed_loop(ctx, g["q0"][i][None], str(motion), g["amplitude"][i:i + 1], g["phase"][i:i + 1],
                                  ticks=g["u"].shape[1], solver=solver)
        finally:
            if solver is not None:
                solver.close()
        assert got.converged.all()
        np.testing.assert_allclose(got.u[:, 0], g["u"][i], rtol=0, atol=SOL_ATOL)
        np.testing.assert_allclose(got.q[:, 0], g["q"][i], rtol=0, atol=SOL_ATOL)


def test_closed_loop_survives_a_tick_on_which_every_controller_exits_early():
    """Equal platform phases: at the velocity zero-crossing EVERY controller takes the early exit (ur_mpc.py:108),
    after ticks that solved.  The loop must go on (zero command, nothing solved) and solve again afterwards."""
    from mpc_amd.closed_loop import run_closed_loop
    from mpc_amd.scenarios import PLATFORM_FREQ_HZ, Q_LINEAR_READY

    mp = MpcParams.reference_yaml_defaults(horizon=1)
    ctx = MpcContext(mp, 2, 0)
    w = 2 * np.pi * PLATFORM_FREQ_HZ
    phase = np.full(2, np.pi / 2 - 2 * w * mp.dt)           # A w cos(w t + phase) = 0 at tick 2
    q0 = np.stack([Q_LINEAR_READY, Q_LINEAR_READY + 0.05])
    got = run_closed_loop(ctx, q0, "linear", np.array([0.2, 0.2]), phase, ticks=5)
    assert np.array_equal(got.u[2], np.zeros((2, 6)))        # the early-exit tick
    assert np.abs(got.u[[0, 1, 3, 4]]).max() > 1e-4 and got.converged.all()
    np.testing.assert_array_equal(got.q[3], got.q[2])


def test_full_size_solves_65536_angular():
    """configs[2] at full size through DeviceQpSLSQP and ResidentSLSQP: 65,536 angular-motion scenarios, H=20, one tick.
    Scenarios are independent optimisations, so (i) a permuted batch gives the permuted results bit for bit,
    (ii) a half batch equals that half of the full batch, (iii) everything past the early exit converges, and
    (iv) a strided sample equals the host-only driver (which is pinned call by call to scipy's core)."""
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.native_solver import DeviceQpSLSQP, NativeBatchedSLSQP
    from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    N = 65536
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    ctx = MpcContext(mp, N, 0)
    hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3, goal_jitter=0.0)
    keep = np.nonzero(hs.active)[0]                          # compute_control solves only these (ur_mpc.py:108)
    assert len(keep) > 60000
    sc_all, _ = to_device(ctx, hs)
    x0 = np.tile(hs.u_init, (1, mp.horizon))                 # the cold start of compute_control (ur_mpc.py:183-185)
    gather = lambda rows: ctx.gather(sc_all, torch.from_numpy(rows.astype(np.int32)).to(ctx.device),
                                     __import__("mpc_amd.batch", fromlist=["ScenarioBatch"]).ScenarioBatch.empty(
                                         ctx.device, len(rows), 20, False))
    with DeviceQpSLSQP(ctx, threads=12, capacity=N) as dev:
        full = dev.solve(gather(keep), x0[keep])
        assert full.success.all() and full.n_host_fallback == 0
        perm = np.random.default_rng(1).permutation(len(keep))
        shuffled = dev.solve(gather(keep[perm]), x0[keep[perm]])
        np.testing.assert_array_equal(shuffled.x, full.x[perm])
        np.testing.assert_array_equal(shuffled.nit, full.nit[perm])
        np.testing.assert_array_equal(shuffled.fun, full.fun[perm])
        h = len(keep) // 2
        half = dev.solve(gather(keep[h:]), x0[keep[h:]])
        np.testing.assert_array_equal(half.x, full.x[h:])
        np.testing.assert_array_equal(half.status, full.status[h:])
    with ResidentSLSQP(ctx, capacity=N) as res:              # the driver bench.py times: everything on the device
        whole = res.solve(gather(keep), x0[keep])
    np.testing.assert_array_equal(whole.x, full.x)
    np.testing.assert_array_equal(whole.nit, full.nit)
    np.testing.assert_array_equal(whole.status, full.status)
    np.testing.assert_array_equal(whole.fun, full.fun)
    assert (whole.n_eval, whole.n_eval_fd, whole.n_qp) == (full.n_eval, full.n_eval_fd, full.n_qp)
    # five QPs in six end with lsq's free step (no transformed matrix, no NNLS): the rows below, solved by the host core
    # (Kraft's LSQ as scipy has it, no such shortcut), say that those steps are the long form's bit for bit
    assert whole.n_free_steps > 0.7 * whole.n_qp
    sample = np.arange(0, len(keep), 251)
    with NativeBatchedSLSQP(ctx, threads=12, capacity=len(sample)) as host:
        want = host.solve(gather(keep[sample]), x0[keep[sample]])
    np.testing.assert_array_equal(full.x[sample], want.x)
    np.testing.assert_array_equal(full.nit[sample], want.nit)
    np.testing.assert_array_equal(full.status[sample], want.status)
    lim = np.tile(np.asarray(mp.v_lim), mp.horizon)
    assert np.all(np.abs(full.x) <= lim + 1e-12)             # every converged sequence respects the bounds


def test_benchmark_solves_reproduce_the_long_form_hash():
    """The 62,838 solves of the benchmark's solves leg (BASELINE configs[2]: 65,536 angular-motion scenarios, those past
    the early exit) with the resident driver: a SHA-256 over every solution, f, iteration count and exit mode.  The value
    below was computed with the library as it was BEFORE the QP kernel learned to try the unconstrained step first
    (commit bbd06ce: every QP through the row transform and NNLS) and has been reproduced by every build since
    (tools/resident_time.py --hash; profiles/r03/experiments/README.md): 1,137,720 QPs, five in six of them now
    answered by the free step, not one bit different."""
    import hashlib

    from mpc_amd.batch import MpcContext, ScenarioBatch
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

    N = 65536
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    ctx = MpcContext(mp, N, 0)
    hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3, goal_jitter=0.0)
    keep = np.nonzero(hs.active)[0]
    sc_all, _ = to_device(ctx, hs)
    sc = ctx.gather(sc_all, torch.from_numpy(keep.astype(np.int32)).to(ctx.device),
                    ScenarioBatch.empty(ctx.device, len(keep), 20, False))
    x0 = np.tile(hs.u_init, (1, 20))[keep]
    with ResidentSLSQP(ctx, capacity=len(keep), lanes=1) as dev:
        r = dev.solve(sc, x0)
    assert len(keep) == 62838 and int(r.nit.sum()) == 1137720 and r.success.all()
    h = hashlib.sha256()
    for a in (r.x, r.fun, r.nit.astype(np.int64), r.status.astype(np.int64)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == "df4ca96c8f3255e4a91f489b241c8a948b09cb998d69d185bb76edf6fd1d39d7"
    assert r.n_free_steps > 0.8 * r.n_qp

