"""GPU: the drop-in GpuURMPC against full compute_control ticks of the real reference (golden)."""
import os

import numpy as np
import pytest
import torch

from mpc_amd.batch import MpcContext
from mpc_amd.params import MpcParams
from mpc_amd.ur_mpc import GpuURMPC

pytestmark = pytest.mark.gpu

SOL_ATOL = 1e-6   # north star: converged joint trajectory within 1e-6 rad


def _dicts(g, i):
    js = {"position": g["q0"][i].copy(), "velocity": np.zeros(6)}
    ee = {"position": g["ee"][i, :3].copy(), "orientation": g["ee"][i, 3:].copy()}
    tg = {"position": g["target"][i, :3].copy(), "orientation": g["target"][i, 3:].copy()}
    b = g["base"][i]
    bs = {"position": b[0:3].copy(), "orientation": b[3:6].copy(), "linear_velocity": b[6:9].copy(),
          "angular_velocity": b[9:12].copy()}
    return js, ee, tg, bs


def test_drop_in_class_with_the_resident_solver_equals_scipy_minimize():
    """solver="resident" replaces nothing but scipy's minimize: same previous_solution and control as the default
    class, bit for bit, over warm-started ticks (H = 3 and 20), and the same early exit."""
    for H in (3, 20):
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"solve_H{H}.npz"), allow_pickle=False)
        a = GpuURMPC(MpcParams.reference_yaml_defaults(horizon=H))
        b = GpuURMPC(MpcParams.reference_yaml_defaults(horizon=H), solver="resident")
        js, ee, tg, bs = _dicts(g, 0)
        for tick in range(3):
            ua, ub = a.compute_control(js, ee, tg, bs), b.compute_control(js, ee, tg, bs)
            assert np.array_equal(ua, ub) and np.array_equal(a.previous_solution, b.previous_solution)
            assert b.last_result.success and b.last_result.nit == a.last_result.nit
    with pytest.raises(ValueError):
        GpuURMPC(solver="newton")


@pytest.mark.parametrize("H", [1, 3, 20])
def test_compute_control_matches_reference(golden_dir, H):
    path = os.path.join(golden_dir, f"solve_H{H}.npz")
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    g = np.load(path, allow_pickle=False)
    for i in range(g["q0"].shape[0]):
        mpc = GpuURMPC(MpcParams.reference_yaml_defaults(horizon=H))
        js, ee, tg, bs = _dicts(g, i)
        for tick in range(g["control"].shape[1]):   # tick 1 exercises warm start, damping, EMA
            u = mpc.compute_control(js, ee, tg, bs)
            np.testing.assert_allclose(mpc.previous_solution, g["solution"][i, tick], rtol=0, atol=SOL_ATOL)
            np.testing.assert_allclose(u, g["control"][i, tick], rtol=0, atol=SOL_ATOL)


@pytest.mark.parametrize("solver", ["scipy", "resident"])
def test_commanded_velocities_reach_the_platform_model(golden_dir, solver):
    """A platform state with 'v_cmd' / 'omega_cmd' (ur_mpc.py:624-625) through the device predictor and through whole
    compute_control ticks, against what the real reference made of the same states (base_motion_cmd.npz)."""
    path = os.path.join(golden_dir, "base_motion_cmd.npz")
    if not os.path.exists(path):
        pytest.fail("base_motion_cmd.npz missing: python -m oracle.gen_golden --base-cmd (where /root/reference exists)")
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    g = np.load(path, allow_pickle=False)
    H = int(g["H"])
    for i in range(g["base"].shape[0]):
        b = g["base"][i]
        bs = {"position": b[0:3].copy(), "orientation": b[3:6].copy(), "linear_velocity": b[6:9].copy(),
              "angular_velocity": b[9:12].copy()}
        if not np.isnan(g["cmd"][i, 0]):
            bs["v_cmd"] = float(g["cmd"][i, 0])
        if not np.isnan(g["cmd"][i, 1]):
            bs["omega_cmd"] = float(g["cmd"][i, 1])
        mpc = GpuURMPC(MpcParams.reference_yaml_defaults(horizon=H), solver=solver)
        got = mpc.predict_base_motion(bs, H)
        arr = np.array([np.concatenate([e["position"], e["orientation"], e["linear_velocity"], e["angular_velocity"]])
                        for e in got])
        np.testing.assert_allclose(arr, g["dynamic"][i], rtol=1e-14, atol=1e-15)     # the device's sin/cos vs numpy's
        js = {"position": g["q0"][i].copy(), "velocity": np.zeros(6)}
        ee = {"position": g["ee"][i, :3].copy(), "orientation": g["ee"][i, 3:].copy()}
        tg = {"position": g["target"][i, :3].copy(), "orientation": g["target"][i, 3:].copy()}
        for tick in range(2):
            u = mpc.compute_control(js, ee, tg, bs)
            assert np.abs(u).max() > 1e-3                                            # not the zero command of a swallowed error
            np.testing.assert_allclose(mpc.previous_solution, g["solution"][i, tick], rtol=0, atol=SOL_ATOL)
            np.testing.assert_allclose(u, g["control"][i, tick], rtol=0, atol=SOL_ATOL)


def test_interface_matches_reference_class():
    mpc = GpuURMPC(MpcParams.reference_yaml_defaults(horizon=3))
    assert mpc.dt == 0.07 and mpc.horizon == 3 and mpc.dynamics.n_controls == 6
    assert not hasattr(mpc, "previous_control") and mpc.previous_solution is None
    bounds = mpc._get_bounds(np.zeros(18))
    assert len(bounds) == 18 and bounds[2] == (-2.617993877991494, 2.617993877991494)
    # early exit: platform at rest -> zero command, nothing launched (ur_mpc.py:108-109)
    z = np.zeros(3)
    u = mpc.compute_control({"position": np.zeros(6), "velocity": np.zeros(6)}, {"position": z, "orientation": z},
                            {"position": z, "orientation": z},
                            {"position": z, "orientation": z, "linear_velocity": z, "angular_velocity": z})
    assert np.array_equal(u, np.zeros(6))
    # constraint closure and predict_trajectory agree with each other
    q0 = np.array([5.15, -0.79, -2.24, -0.95, 1.25, 3.45])
    pose = mpc.dynamics.forward_kinematics(q0)
    s0 = np.concatenate([q0, np.zeros(6), pose["position"], pose["orientation"]])
    bt = mpc.predict_base_motion({"position": z, "orientation": z, "linear_velocity": np.array([0.1, 0, 0]),
                                  "angular_velocity": z}, 3)
    x = np.linspace(-0.2, 0.2, 18)
    c = mpc._get_constraints(s0, bt)[0]["fun"](x)
    traj = mpc.predict_trajectory(s0, x.reshape(3, 6), 3)
    lim = np.array(list(mpc.dynamics.joint_vel_limits.values()))
    want = np.concatenate([np.concatenate([np.minimum(s[:6] + 2 * np.pi, 2 * np.pi - s[:6]), lim - np.abs(s[6:12])])
                           for s in traj])
    np.testing.assert_allclose(c, want, rtol=0, atol=1e-13)


@pytest.mark.parametrize("H", [1, 3, 20])
def test_batched_mpc_matches_reference(golden_dir, H):
    """All golden scenarios of one horizon advance together through the lock-step SLSQP driver."""
    from mpc_amd.batch import MpcContext
    from mpc_amd.solver import BatchedMPC

    path = os.path.join(golden_dir, f"solve_H{H}.npz")
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path, allow_pickle=False)
    N = g["q0"].shape[0]
    ctx = MpcContext(MpcParams.reference_yaml_defaults(horizon=H), max_scenarios=N, device=0)
    bm = BatchedMPC(ctx)
    for tick in range(g["control"].shape[1]):
        u = bm.compute_control(g["q0"], g["ee"], g["target"], g["base"])
        assert bm.last.success.all()
        np.testing.assert_allclose(bm.previous_solution, g["solution"][:, tick], rtol=0, atol=SOL_ATOL)
        np.testing.assert_allclose(u, g["control"][:, tick], rtol=0, atol=SOL_ATOL)


def test_batched_solver_equals_single_scenario_driver():
    """64 seeded scenarios: the lock-step driver and scipy.minimize with device callbacks agree."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.scenarios import to_device
    from mpc_amd.solver import BatchedSLSQP
    from scipy.optimize import minimize

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    _, hs = host_scenarios(mp, 64, motion="angular", cfg_id=3)
    ctx = MpcContext(mp, 64, 0)
    sc, _ = to_device(ctx, hs)
    res = BatchedSLSQP(ctx).solve(sc, hs.x)
    assert res.success.sum() >= 60
    one = MpcContext(mp, 1, 0)
    for i in (0, 17, 63):
        sci = sc.slice(i, i + 1)

        def ev(x):
            r = one.eval_fd(sci, torch.as_tensor(np.clip(x, -np.tile(mp.v_lim, 3), np.tile(mp.v_lim, 3)).reshape(1, -1)).to(one.device))
            return float(r.cost.cpu()[0]), r.grad[0].cpu().numpy(), r.con[0].cpu().numpy(), r.con_jac[0].cpu().numpy().T

        ref = minimize(lambda x: ev(x)[0], hs.x[i], jac=lambda x: ev(x)[1], method="SLSQP", bounds=mp.bounds(),
                       constraints=[{"type": "ineq", "fun": lambda x: ev(x)[2], "jac": lambda x: ev(x)[3]}],
                       options={"ftol": 1e-3, "maxiter": 50})
        np.testing.assert_allclose(res.x[i], ref.x, rtol=0, atol=1e-12)
        assert res.nit[i] == ref.nit and res.status[i] == ref.status


@pytest.mark.parametrize("motion,H", [("linear", 3), ("angular", 3), ("linear", 1)])
def test_closed_loop_trajectory_matches_reference_path(motion, H):
    """Six ticks of the headless loop: the joint trajectory of the batched GPU controllers stays within
    1e-6 rad of the numpy restatement of the reference driven through the same loop (which is
    bit-identical to the reference's compute_control, tests/test_oracle.py)."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.closed_loop import run_closed_loop
    from mpc_amd.scenarios import platform_state, world_ee_pose
    from oracle.numpy_ref import NumpyRefMPC, ur20_fk
    from oracle.oracle import ref_params_from

    mp = MpcParams.reference_yaml_defaults(horizon=H)
    _, hs = host_scenarios(mp, 4, motion=motion, cfg_id=7)
    ticks = 6
    ctx = MpcContext(mp, 4, 0)
    got = run_closed_loop(ctx, hs.q0, motion, hs.amplitude, hs.phase, ticks)
    for i in range(4):
        ref = NumpyRefMPC(ref_params_from(mp))
        q = hs.q0[i].copy()
        fk = lambda qq, b: world_ee_pose(np.concatenate(ur20_fk(qq))[None], b)[0]
        base = platform_state(motion, hs.amplitude[i], hs.phase[i], 0.0)
        ee = fk(q, base)
        target = {"position": ee[:3].copy(), "orientation": ee[3:].copy()}
        for t in range(ticks):
            b = platform_state(motion, hs.amplitude[i], hs.phase[i], t * mp.dt)[0]
            bs = {"position": b[0:3], "orientation": b[3:6], "linear_velocity": b[6:9], "angular_velocity": b[9:12]}
            u = ref.compute_control({"position": q, "velocity": np.zeros(6)},
                                    {"position": ee[:3], "orientation": ee[3:]}, target, bs)
            q = q + u * mp.dt
            ee = fk(q, platform_state(motion, hs.amplitude[i], hs.phase[i], (t + 1) * mp.dt))
            np.testing.assert_allclose(got.q[t + 1, i], q, rtol=0, atol=SOL_ATOL)


def test_solver_pool_equals_single_process():
    """Worker processes running the SLSQP cores, this process serving their evaluations from the GPU:
    same iterates, exit modes and iteration counts as the one-process batch, row for row."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.pool import SolverPool
    from mpc_amd.scenarios import to_device
    from mpc_amd.solver import BatchedSLSQP

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    ctx = MpcContext(mp, 64, 0)
    with SolverPool(ctx, workers=3, capacity=40) as pool:
        for count, with_prev in ((21, False), (40, True), (2, False)):      # 2 scenarios: one worker idle
            _, hs = host_scenarios(mp, count, motion="linear", cfg_id=8, with_prev=with_prev)
            sc, _ = to_device(ctx, hs)
            one = BatchedSLSQP(ctx).solve(sc, hs.x)
            many = pool.solve(sc, hs.x)
            np.testing.assert_array_equal(many.x, one.x)
            np.testing.assert_array_equal(many.status, one.status)
            np.testing.assert_array_equal(many.nit, one.nit)
            np.testing.assert_array_equal(many.fun, one.fun)
            assert many.n_eval == one.n_eval and many.n_eval_fd == one.n_eval_fd
        with pytest.raises(ValueError):
            _, hs = host_scenarios(mp, 41, motion="linear", cfg_id=8)
            pool.solve(to_device(ctx, hs)[0], hs.x)


def test_prepare_tick_matches_host_assembly():
    """mpc_prepare_tick (one launch) against the pieces compute_control assembles on the host: exact for
    state0, the platform prediction, Jc, the early-exit flag and the shape of x0; the 6x6 DLS solve
    agrees with LAPACK's to rounding."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.scenarios import dls_guess

    mp = MpcParams.reference_yaml_defaults(horizon=20)
    N = 257
    ctx = MpcContext(mp, N, 0)
    _, hs = host_scenarios(mp, N, motion="angular", cfg_id=12)
    base = hs.base.copy()
    base[::7, 6:12] = 0.0                      # these controllers take the early exit
    base[1::7, 11] = 0.009                     # below the threshold too
    ee = hs.state0[:, 12:18]
    rng = np.random.default_rng(5)
    prev = rng.uniform(-0.3, 0.3, (N, mp.n))
    has = (np.arange(N) % 2).astype(np.uint8)
    up = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(ctx.device)
    sc, u_init, x0, active = ctx.prepare_tick(up(hs.q0), up(ee), up(base), up(hs.target), prev_solution=up(prev),
                                              has_solution=up(has, torch.uint8))
    torch.cuda.synchronize()
    want_active = ~((np.linalg.norm(base[:, 6:9], axis=1) < 0.01) & (np.linalg.norm(base[:, 9:12], axis=1) < 0.01))
    np.testing.assert_array_equal(active.cpu().numpy().astype(bool), want_active)
    assert not want_active[0] and not want_active[1] and want_active[2]
    np.testing.assert_array_equal(sc.state0.cpu().numpy(), np.concatenate([hs.q0, np.zeros((N, 6)), ee], axis=1))
    assert torch.equal(sc.base_traj, ctx.predict_base_dynamic(up(base)))
    assert torch.equal(sc.Jc.view(N, 6, 6), ctx.jacobian(up(hs.q0)))
    u_host = dls_guess(sc.Jc.cpu().numpy(), base, ee[:, :3])
    u_dev = u_init.cpu().numpy()
    np.testing.assert_allclose(u_dev, u_host, rtol=1e-9, atol=1e-13)
    want_x0 = np.tile(u_dev, (1, mp.horizon))
    rolled = np.roll(prev, -6, axis=1)
    rolled[:, -6:] = u_dev
    want_x0[has == 1] = rolled[has == 1]
    np.testing.assert_array_equal(x0.cpu().numpy(), want_x0)
    # no previous solution at all: x0 is the guess repeated
    _, u2, x02, _ = ctx.prepare_tick(up(hs.q0), up(ee), up(base), up(hs.target))
    assert torch.equal(u2, u_init) and torch.equal(x02, u_init.repeat(1, mp.horizon))


def test_batched_ticks_with_device_preparation_match_host_preparation():
    """Three ticks of 12 controllers: the start points made on the device lead to the same controls
    (1e-6 rad/s) and the same exit modes as the LAPACK-made ones."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.solver import BatchedMPC

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    N = 12
    ctx = MpcContext(mp, N, 0)
    _, hs = host_scenarios(mp, N, motion="linear", cfg_id=13)
    a, b = BatchedMPC(ctx, prep="host"), BatchedMPC(ctx, prep="device")
    q, ee = hs.q0.copy(), hs.state0[:, 12:18].copy()
    for tick in range(3):
        ua = a.compute_control(q, ee, hs.target, hs.base)
        ub = b.compute_control(q, ee, hs.target, hs.base)
        np.testing.assert_allclose(ub, ua, rtol=0, atol=1e-6)
        np.testing.assert_array_equal(b.last.status, a.last.status)
        q = q + ua * mp.dt


def test_resident_ticks_keep_start_points_and_solutions_on_the_device():
    """prep="resident": mpc_prepare_tick's start points go straight into the device-resident solve and the solutions
    stay on the device as the next tick's warm start; only the first controls come back.  Same controls, exit modes
    and previous solutions, bit for bit, as prep="device" with the same solver (which moves both through the host),
    over four ticks with two controllers idle (early exit) in the first two."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.solver import BatchedMPC

    for H, N in ((3, 12), (20, 8)):
        mp = MpcParams.reference_yaml_defaults(horizon=H)
        ctx = MpcContext(mp, N, 0)
        _, hs = host_scenarios(mp, N, motion="angular", cfg_id=31, goal_jitter=0.0 if H == 20 else 0.05)
        with ResidentSLSQP(ctx) as sa, ResidentSLSQP(ctx) as sb:
            a, b = BatchedMPC(ctx, solver=sa, prep="device"), BatchedMPC(ctx, solver=sb, prep="resident")
            with pytest.raises(ValueError):
                BatchedMPC(ctx, prep="resident")                     # the default solver works through the host
            q, ee = hs.q0.copy(), hs.state0[:, 12:18].copy()
            for tick in range(4):
                base = hs.base.copy()
                if tick < 2:
                    base[[1, 5], 6:12] = 0.0                         # platform at rest: ur_mpc.py:108 returns zeros
                ua = a.compute_control(q, ee, hs.target, base)
                ub = b.compute_control(q, ee, hs.target, base)
                np.testing.assert_array_equal(ub, ua)
                np.testing.assert_array_equal(b.last.status, a.last.status)
                np.testing.assert_array_equal(b.last.nit, a.last.nit)
                np.testing.assert_array_equal(b.last_rows, a.last_rows)
                assert b.last.x is None and a.last.x is not None
                np.testing.assert_array_equal(b.previous_solution, a.previous_solution)
                q = q + ua * mp.dt
            assert np.isnan(a.previous_solution).sum() == 0


def test_closed_loop_with_the_resident_tick_matches_the_reference_trajectory(golden_dir):
    """The action server's loop (mpc_action_server.py:77-135) with everything between a tick's inputs and its commands
    on the device (prep="resident"): four ticks per platform family against the REAL reference's closed-loop
    trajectories (tests/golden/closed_loop_H20.npz) at 1e-6 rad, and equal to the prep="device" loop bit for bit."""
    import os

    from mpc_amd.batch import MpcContext
    from mpc_amd.closed_loop import run_closed_loop
    from mpc_amd.device_solver import ResidentSLSQP

    g = np.load(os.path.join(golden_dir, "closed_loop_H20.npz"), allow_pickle=False)
    mp = MpcParams.reference_yaml_defaults(horizon=int(g["H"]))
    ctx = MpcContext(mp, 4, 0)
    for i, motion in enumerate(g["motion"]):
        q0 = g["q0"][i][None]
        amp, ph = np.array([float(g["amplitude"][i])]), np.array([float(g["phase"][i])])
        ticks = g["u"][i].shape[0]
        with ResidentSLSQP(ctx) as solver:
            res = run_closed_loop(ctx, q0, str(motion), amp, ph, ticks=ticks, solver=solver, prep="resident")
            dev = run_closed_loop(ctx, q0, str(motion), amp, ph, ticks=ticks, solver=solver, prep="device")
        assert res.converged.all()
        np.testing.assert_allclose(res.q[:, 0], g["q"][i], rtol=0, atol=1e-6)
        np.testing.assert_allclose(res.u[:, 0], g["u"][i], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(res.q, dev.q)
        np.testing.assert_array_equal(res.u, dev.u)


def test_batched_mpc_with_kinematic_predictor_equals_drop_in_controllers():
    """use_dynamic_model = False (the reference's code default, ur_mpc.py:63): the batched tick with the
    batched kinematic predictor against one drop-in GpuURMPC per controller, three ticks with a moving
    platform (the predictor's filters carry state from tick to tick)."""
    import dataclasses

    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.solver import BatchedMPC

    mp = dataclasses.replace(MpcParams.reference_yaml_defaults(horizon=3), use_dynamic_model=False)
    N = 6
    _, hs = host_scenarios(mp, N, motion="linear", cfg_id=14)
    ctx = MpcContext(mp, N, 0)
    bm = BatchedMPC(ctx)
    singles = [GpuURMPC(mp) for _ in range(N)]
    ee = hs.state0[:, 12:18]
    for tick in range(3):
        base = hs.base.copy()
        base[:, 6] *= 1.0 + 0.1 * tick            # the platform speeds up: non-zero filtered acceleration
        base[:, 0] += 0.01 * tick
        u = bm.compute_control(hs.q0, ee, hs.target, base)
        for i in range(N):
            ui = singles[i].compute_control(
                {"position": hs.q0[i], "velocity": np.zeros(6)}, {"position": ee[i, :3], "orientation": ee[i, 3:]},
                {"position": hs.target[i, :3], "orientation": hs.target[i, 3:]},
                {"position": base[i, 0:3], "orientation": base[i, 3:6], "linear_velocity": base[i, 6:9],
                 "angular_velocity": base[i, 9:12]})
            np.testing.assert_allclose(u[i], ui, rtol=0, atol=1e-9)


def test_solver_pool_reports_a_dead_worker():
    """A worker that dies mid-solve must surface as an error, not as a hang; the pool closes itself."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.pool import SolverPool
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=20)
    ctx = MpcContext(mp, 32, 0)
    _, hs = host_scenarios(mp, 32, motion="angular", cfg_id=15)
    sc, _ = to_device(ctx, hs)
    pool = SolverPool(ctx, workers=2, capacity=32)
    try:
        victim = pool._procs[1]
        killed = []

        def kill_once(t, nf, ng, done):
            if not killed:
                victim.kill()
                killed.append(True)

        with pytest.raises(RuntimeError, match="worker"):
            pool.solve(sc, hs.x, progress=kill_once, progress_every=0.0)
        with pytest.raises(RuntimeError, match="closed"):
            pool.solve(sc, hs.x)
    finally:
        pool.close()


def test_closed_loop_through_the_pool_equals_one_process():
    """Six ticks of 24 controllers: worker processes + device-side bookkeeping change nothing in the trajectory."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.closed_loop import run_closed_loop
    from mpc_amd.pool import SolverPool

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    N = 24
    ctx = MpcContext(mp, N, 0)
    _, hs = host_scenarios(mp, N, motion="angular", cfg_id=16)
    one = run_closed_loop(ctx, hs.q0, "angular", hs.amplitude, hs.phase, ticks=6)
    with SolverPool(ctx, workers=3, capacity=N) as pool:
        many = run_closed_loop(ctx, hs.q0, "angular", hs.amplitude, hs.phase, ticks=6, solver=pool)
    np.testing.assert_array_equal(many.q, one.q)
    np.testing.assert_array_equal(many.u, one.u)
    np.testing.assert_array_equal(many.converged, one.converged)


def test_device_qp_driver_hands_inconsistent_linearisations_to_the_host():
    """A joint that starts beyond its limit makes a constraint row negative with a zero gradient: the linearised QP
    is inconsistent (mode 4) and Kraft's body switches to the augmented problem, which the device does not solve.
    Such problems are marked and solved again by the host-only driver; the rest of the batch is untouched and every
    result equals the host-only driver's."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.native_solver import DeviceQpSLSQP, NativeBatchedSLSQP
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    N = 12
    ctx = MpcContext(mp, N, 0)
    _, hs = host_scenarios(mp, N, motion="angular", cfg_id=23)
    bad = [2, 7]
    hs.state0[bad, 0] = mp.q_limit + 0.2
    sc, _ = to_device(ctx, hs)
    with NativeBatchedSLSQP(ctx, threads=2) as host:
        one = host.solve(sc, hs.x)
    with DeviceQpSLSQP(ctx, threads=2) as dev:
        two = dev.solve(sc, hs.x)
    assert two.n_host_fallback == len(bad)
    np.testing.assert_array_equal(two.status, one.status)
    np.testing.assert_array_equal(two.nit, one.nit)
    np.testing.assert_array_equal(two.x, one.x)
    np.testing.assert_array_equal(two.fun, one.fun)


def test_closed_loop_with_the_qp_on_the_device_equals_one_process():
    """Six ticks of 24 controllers at H=20 and of 160 at H=3 (two lanes): with the BFGS update and the QP of every
    iteration on the GPU the trajectories are those of scipy's own core, bit for bit (warm starts, EMA, clip
    included)."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.closed_loop import run_closed_loop
    from mpc_amd.native_solver import DeviceQpSLSQP

    for H, N, ticks in ((20, 24, 4), (3, 160, 6)):
        mp = MpcParams.reference_yaml_defaults(horizon=H)
        ctx = MpcContext(mp, N, 0)
        _, hs = host_scenarios(mp, N, motion="angular", cfg_id=16)
        one = run_closed_loop(ctx, hs.q0, "angular", hs.amplitude, hs.phase, ticks=ticks)
        with DeviceQpSLSQP(ctx, threads=3, capacity=N, lanes=2, min_per_lane=64) as dev:
            many = run_closed_loop(ctx, hs.q0, "angular", hs.amplitude, hs.phase, ticks=ticks, solver=dev)
        np.testing.assert_array_equal(many.q, one.q)
        np.testing.assert_array_equal(many.u, one.u)
        np.testing.assert_array_equal(many.converged, one.converged)


def test_native_driver_equals_scipy_core_driver():
    """libmpc_slsqp.so (C core on host threads, evaluations from the GPU) against the lock-step driver over
    scipy's compiled core: identical iterates, exit modes, iteration counts and evaluation counts."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.native_solver import NativeBatchedSLSQP
    from mpc_amd.scenarios import to_device
    from mpc_amd.solver import BatchedMPC, BatchedSLSQP

    for H, count, prev in ((3, 40, True), (20, 24, False)):
        mp = MpcParams.reference_yaml_defaults(horizon=H)
        ctx = MpcContext(mp, 64, 0)
        _, hs = host_scenarios(mp, count, motion="angular", cfg_id=18, with_prev=prev)
        sc, _ = to_device(ctx, hs)
        one = BatchedSLSQP(ctx).solve(sc, hs.x)
        with NativeBatchedSLSQP(ctx, threads=3) as native:
            two = native.solve(sc, hs.x)
            again = native.solve(sc.slice(0, 5), hs.x[:5])           # the batch object is reusable
            np.testing.assert_array_equal(two.x, one.x)
            np.testing.assert_array_equal(two.status, one.status)
            np.testing.assert_array_equal(two.nit, one.nit)
            np.testing.assert_array_equal(two.fun, one.fun)
            assert (two.n_eval, two.n_eval_fd) == (one.n_eval, one.n_eval_fd)
            np.testing.assert_array_equal(again.x, one.x[:5])
            # and as the solver of the batched controller
            a, b = BatchedMPC(ctx), BatchedMPC(ctx, solver=native)
            for _ in range(2):
                ua = a.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
                ub = b.compute_control(hs.q0, hs.state0[:, 12:18], hs.target, hs.base)
                np.testing.assert_array_equal(ub, ua)


@pytest.mark.parametrize("H,N", [(1, 6), (3, 6), (20, 3)])
def test_qp_subproblem_on_the_device_is_bit_identical_to_the_host_core(H, N):
    """Every QP subproblem that whole host solves pose (recorded through the core's LSQ hook: LDL' factor,
    gradient, constraint values, structured dc/dx, box) is solved again by mpc_qp_solve, one workgroup per
    problem: same step, same constraint multipliers, same mode, bit for bit."""
    import ctypes

    from helpers import host_scenarios
    from mpc_amd import native_solver
    from mpc_amd.batch import MpcContext
    from mpc_amd.native_solver import NativeBatch
    from mpc_amd.slsqp_state import nz_scatter_index

    lib = native_solver.load()
    mp = MpcParams.reference_yaml_defaults(horizon=H)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=43, goal_jitter=0.0 if H == 20 else 0.05)
    n, m = mp.n, mp.m
    lb = np.array([b[0] for b in mp.bounds()])
    ub = np.array([b[1] for b in mp.bounds()])
    scatter = nz_scatter_index(H)
    ldl = (n + 1) * n // 2 + 1
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    FN = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, dp, dp,
                          dp, dp, dp, ip, ip)
    builtin = FN(("mpc_slsqp_lsq_builtin", lib))
    rec = []

    def hook(m_, meq, n_, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode):
        arr = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
        inp = None
        if n_ == n and nl == ldl:
            inp = (arr(l, ldl), arr(g, n), arr(b, m_), arr(a, (n + 1) * la)[scatter], arr(xl, n), arr(xu, n))
            if len(rec) < 4 * N:
                # The same problem with operands far outside the range in which the kernel's single-wavefront
                # recurrences (csrc/lsq_wave.hpp) split their divisions: a tiny gradient, a zero gradient (forward
                # substitution declines), huge constraint values (back substitution declines, having restored its
                # input).  The team code must take over with the same bits as ever.
                for gs, bs in ((2.0 ** -460, 1.0), (0.0, 1.0), (1.0, 2.0 ** 450)):
                    g2, b2 = inp[1] * gs, inp[2] * bs
                    x2, y2, md2 = np.zeros(n), np.zeros(m_ + 2 * n + 2), ctypes.c_int(0)
                    builtin(m_, meq, n_, nl, la, l, g2.ctypes.data_as(dp), a, b2.ctypes.data_as(dp), xl, xu,
                            x2.ctypes.data_as(dp), y2.ctypes.data_as(dp), w, jw, ctypes.byref(md2))
                    rec.append((inp[0], g2, b2, inp[3], inp[4], inp[5], x2.copy(), y2[:m_].copy(), md2.value))
        builtin(m_, meq, n_, nl, la, l, g, a, b, xl, xu, x, y, w, jw, mode)
        if inp is not None:
            rec.append(inp + (arr(x, n), arr(y, m_), mode[0]))

    cb = FN(hook)
    lib.mpc_slsqp_set_lsq(ctypes.cast(cb, ctypes.c_void_p))
    try:
        def rows(idx, X):
            d = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
            d["x"] = np.ascontiguousarray(X)
            return d
        b = NativeBatch(H, N, threads=1)
        b.begin(np.tile(hs.u_init, (1, H)), lb, ub, mp.ftol, mp.maxiter)
        f, g, c, a = co.eval_fd(rows(np.arange(N), b.X[:N]))
        b.F[:N], b.Cc[:N], b.G[:N, :n], b.NZ[:N] = f, c, g, a.reshape(N, -1)[:, scatter]
        while True:
            wf, wg = (w.copy() for w in b.advance())
            if len(wf) + len(wg) == 0:
                break
            if len(wf):
                f, c = co.eval(rows(wf, np.clip(b.X[wf], lb, ub)))
                b.F[wf], b.Cc[wf] = f, c
            if len(wg):
                _, g, _, a = co.eval_fd(rows(wg, np.clip(b.X[wg], lb, ub)))
                b.G[wg, :n] = g
                b.NZ[wg] = a.reshape(len(wg), -1)[:, scatter]
        b.close()
    finally:
        lib.mpc_slsqp_set_lsq(None)
    K = len(rec)
    assert K >= N
    ctx = MpcContext(mp, max(K, 8), 0)
    up = lambda k: torch.as_tensor(np.stack([r[k] for r in rec])).to(ctx.device)
    s, y, mode = ctx.qp_solve(up(0), up(1), up(2), up(3), up(4), up(5))
    torch.cuda.synchronize()
    want_mode = np.array([r[8] for r in rec])
    np.testing.assert_array_equal(mode.cpu().numpy(), want_mode)
    ok = want_mode == 1                                     # (a failed QP leaves step and multipliers undefined)
    assert ok.sum() >= N
    np.testing.assert_array_equal(s.cpu().numpy()[ok], np.stack([r[6] for r in rec])[ok])
    np.testing.assert_array_equal(y.cpu().numpy()[ok][:, :m], np.stack([r[7] for r in rec])[ok])


def test_device_qp_driver_equals_host_driver():
    """SLSQP with the BFGS update and the QP of every major iteration on the GPU (mpc_qp_step) and the scalar logic
    on host threads: iterates, exit modes, iteration and evaluation counts identical to the host-only driver."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.native_solver import DeviceQpSLSQP, NativeBatchedSLSQP
    from mpc_amd.scenarios import to_device

    # capacity 256, at least 32 per lane: four lanes of 64; the batches of 150 and 200 run as four parts on four streams,
    # the small ones on one
    # H = 32: the team's triangle and vectors (175 KB) no longer fit LDS and live in its global workspace slice
    for H, count, prev in ((1, 9, False), (3, 200, True), (20, 24, False), (20, 150, False), (32, 6, False)):
        mp = MpcParams.reference_yaml_defaults(horizon=H)
        ctx = MpcContext(mp, 256, 0)
        _, hs = host_scenarios(mp, count, motion="angular", cfg_id=19, with_prev=prev)
        sc, _ = to_device(ctx, hs)
        with NativeBatchedSLSQP(ctx, threads=3) as host:
            one = host.solve(sc, hs.x)
        with DeviceQpSLSQP(ctx, threads=3, min_per_lane=32) as dev:
            assert len(dev.lanes) == 4
            two = dev.solve(sc, hs.x)
            again = dev.solve(sc.slice(0, 5), hs.x[:5])
        case = f"H={H} count={count}"
        np.testing.assert_array_equal(two.status, one.status, err_msg=case)
        np.testing.assert_array_equal(two.nit, one.nit, err_msg=case)
        np.testing.assert_array_equal(two.x, one.x, err_msg=case)
        np.testing.assert_array_equal(two.fun, one.fun, err_msg=case)
        assert (two.n_eval, two.n_eval_fd) == (one.n_eval, one.n_eval_fd), case
        assert two.n_qp >= int(one.nit.sum()), case
        np.testing.assert_array_equal(again.x, one.x[:5], err_msg=case)


def test_resident_driver_equals_host_driver():
    """Whole solves resident on the device (mpc_solve_*: Kraft's body as a kernel, device-side request lists, no
    host SLSQP logic at all): iterates, exit modes, iteration and evaluation counts identical to the host-only
    driver, which is pinned call by call to scipy's compiled core."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.native_solver import NativeBatchedSLSQP
    from mpc_amd.scenarios import to_device

    for H, count, prev in ((1, 9, False), (3, 200, True), (20, 24, False), (20, 150, True), (32, 6, False)):
        mp = MpcParams.reference_yaml_defaults(horizon=H)
        ctx = MpcContext(mp, 256, 0)
        _, hs = host_scenarios(mp, count, motion="angular", cfg_id=19, with_prev=prev)
        sc, _ = to_device(ctx, hs)
        with NativeBatchedSLSQP(ctx, threads=3) as host:
            one = host.solve(sc, hs.x)
        with ResidentSLSQP(ctx, lanes=4, min_per_lane=32, rounds_per_check=5) as dev:
            assert len(dev.lanes) == 4
            two = dev.solve(sc, hs.x)
            again = dev.solve(sc.slice(0, 5), hs.x[:5])
        case = f"H={H} count={count}"
        np.testing.assert_array_equal(two.status, one.status, err_msg=case)
        np.testing.assert_array_equal(two.nit, one.nit, err_msg=case)
        np.testing.assert_array_equal(two.x, one.x, err_msg=case)
        np.testing.assert_array_equal(two.fun, one.fun, err_msg=case)
        assert (two.n_eval, two.n_eval_fd) == (one.n_eval, one.n_eval_fd), case
        assert two.n_qp >= int(one.nit.sum()), case
        np.testing.assert_array_equal(again.x, one.x[:5], err_msg=case)


@pytest.mark.parametrize("H,N,bad", [(3, 12, (2, 7)), (20, 6, (1, 4)), (1, 9, (0, 5, 8))])
def test_resident_driver_solves_inconsistent_linearisations_on_the_device(H, N, bad):
    """A joint that starts beyond its limit makes the linearised QP inconsistent (mode 4): Kraft's body turns to the
    augmented problem (one more variable, weight 100 .. 1e7), which the QP kernel solves as well (lsq<AUG>): nothing
    is handed to the host, and iterates, exit modes (the iteration limit among them) and iteration counts are the
    host-only driver's, i.e. scipy's core's (tests/test_native_slsqp.py pins that driver on the same kind of case)."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.native_solver import NativeBatchedSLSQP
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=H)
    ctx = MpcContext(mp, N, 0)
    _, hs = host_scenarios(mp, N, motion="angular", cfg_id=23, goal_jitter=0.0 if H == 20 else 0.05)
    hs.state0[list(bad), 0] = mp.q_limit + 0.2
    sc, _ = to_device(ctx, hs)
    with NativeBatchedSLSQP(ctx, threads=2) as host:
        one = host.solve(sc, hs.x)
    with ResidentSLSQP(ctx) as dev:
        two = dev.solve(sc, hs.x)
    assert two.n_host_fallback == 0
    np.testing.assert_array_equal(two.status, one.status)
    np.testing.assert_array_equal(two.nit, one.nit)
    np.testing.assert_array_equal(two.x, one.x)
    np.testing.assert_array_equal(two.fun, one.fun)
    assert two.n_qp > int(one.nit.sum())              # the augmented problems were extra requests


@pytest.mark.parametrize("H,N,bad,prev", [(3, 40, (2, 7), True), (20, 24, (), False), (20, 9, (1, 4), False), (1, 9, (0, 5), True),
                                          (20, 700, (), False)])
def test_solves_without_the_lock_step_equal_the_rounds(H, N, bad, prev):
    """mpc_solve_run (one launch, a workgroup per problem at a time: body, evaluations, BFGS update + QP, Kraft's
    augmented problem) against mpc_solve_rounds on the same batch: iterates, exit modes, iteration counts, f and the
    request statistics bit for bit; 700 problems are more than the chip has teams (the queue is walked)."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=H)
    ctx = MpcContext(mp, N, 0)
    _, hs = host_scenarios(mp, N, motion="angular", cfg_id=37, with_prev=prev, goal_jitter=0.0 if H == 20 else 0.05)
    hs.state0[list(bad), 0] = mp.q_limit + 0.2
    sc, _ = to_device(ctx, hs)
    with ResidentSLSQP(ctx, team_max=0) as rounds, ResidentSLSQP(ctx, team_max=N) as team:
        one = rounds.solve(sc, hs.x)
        two = team.solve(sc, hs.x)
        again = team.solve(sc.slice(0, 5), hs.x[:5])
    assert two.rounds == 0 and one.rounds > 0
    np.testing.assert_array_equal(two.status, one.status)
    np.testing.assert_array_equal(two.nit, one.nit)
    np.testing.assert_array_equal(two.x, one.x)
    np.testing.assert_array_equal(two.fun, one.fun)
    assert (two.n_eval, two.n_eval_fd) == (one.n_eval, one.n_eval_fd)
    if not bad:
        assert two.n_qp == one.n_qp                       # (deferred augmented requests are counted again in the rounds)
    # most QPs end with lsq's free step (the unconstrained step satisfies every row with a margin), on both paths alike
    assert two.n_free_steps == one.n_free_steps and one.n_free_steps > (0 if bad else 0.5 * one.nit.sum())
    np.testing.assert_array_equal(again.x, one.x[:5])


@pytest.mark.parametrize("H,N", [(3, 300), (20, 200)])
def test_rounds_on_a_side_stream_of_the_callers_equal_the_default_stream(H, N):
    """mpc_solve_rounds runs a round's evaluations on a stream of the library's own beside the QPs (from 128 unfinished
    problems on), forked from and joined to the caller's stream by events: the same batch solved on torch's default
    stream and on another stream of the caller's gives the same bits, and work queued on the caller's stream right
    behind the solve sees its results."""
    from helpers import host_scenarios
    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=H)
    ctx = MpcContext(mp, N, 0)
    _, hs = host_scenarios(mp, N, motion="angular", cfg_id=41, goal_jitter=0.0 if H == 20 else 0.05)
    sc, _ = to_device(ctx, hs)
    with ResidentSLSQP(ctx, team_max=0) as solver:
        one = solver.solve(sc, hs.x)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=ctx.device)
        with torch.cuda.stream(side):
            two = solver.solve(sc, hs.x, x_on_device=True)
            kept = two.x_dev.clone()                       # queued behind the solve on the same stream
        side.synchronize()
    assert one.rounds > 0
    np.testing.assert_array_equal(two.status, one.status)
    np.testing.assert_array_equal(two.nit, one.nit)
    np.testing.assert_array_equal(two.fun, one.fun)
    np.testing.assert_array_equal(kept.cpu().numpy(), one.x)


def test_solve_run_argument_checks():
    """mpc_solve_run through the C-ABI: it follows mpc_solve_begin directly (after a round it is refused: the request
    lists are then part of the state), and a horizon whose QP vectors do not fit LDS is reported as unsupported, with
    the driver falling back to the rounds by itself."""
    import ctypes as C

    from helpers import host_scenarios
    from mpc_amd import _ffi
    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP, p_supports_team
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    N = 6
    ctx = MpcContext(mp, N, 0)
    lib = ctx._lib
    _, hs = host_scenarios(mp, N, motion="linear", cfg_id=29)
    sc, _ = to_device(ctx, hs)
    scc = ctx._scen_c(sc)
    nbytes = int(lib.mpc_solve_workspace_bytes(3, N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ctx.device)
    x0 = torch.as_tensor(hs.x).to(ctx.device)
    left = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    st = C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)
    run = lambda: lib.mpc_solve_run(ctx._ctx, C.byref(scc), ws.data_ptr(), N, nbytes, 50, left.data_ptr(), st)
    assert run() == -1                                                         # MPC_ERR_INVALID_ARG: no mpc_solve_begin yet
    _ffi.check(ctx._ctx, lib.mpc_solve_begin(ctx._ctx, C.byref(scc), x0.data_ptr(), 1e-3, 50, ws.data_ptr(), N, nbytes, st))
    _ffi.check(ctx._ctx, lib.mpc_solve_rounds(ctx._ctx, C.byref(scc), ws.data_ptr(), N, nbytes, 1, N, left.data_ptr(), st))
    assert run() == -1                                    # a round has run
    _ffi.check(ctx._ctx, lib.mpc_solve_begin(ctx._ctx, C.byref(scc), x0.data_ptr(), 1e-3, 50, ws.data_ptr(), N, nbytes, st))
    assert run() == 0
    torch.cuda.synchronize()
    assert int(left.item()) == 0
    big = MpcParams.reference_yaml_defaults(horizon=32)
    ctx32 = MpcContext(big, 4, 0)
    assert not p_supports_team(ctx32)
    with ResidentSLSQP(ctx32) as solver:
        assert solver.team_max == 0


def test_solve_rounds_reports_a_bound_that_is_too_small_and_leaves_a_finished_solve_alone():
    """mpc_solve_rounds through the C-ABI: `bound` below the number of unfinished problems is reported as -1 (problems
    were left out of a round: the solve is void) instead of a silently short count; `bound` = 0 launches nothing and
    reports the count as it stands; rounds after everything has stopped change nothing."""
    import ctypes as C

    from helpers import host_scenarios
    from mpc_amd import _ffi
    from mpc_amd.batch import MpcContext
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    N = 40
    ctx = MpcContext(mp, N, 0)
    lib = ctx._lib
    _, hs = host_scenarios(mp, N, motion="linear", cfg_id=29)
    sc, _ = to_device(ctx, hs)
    scc = ctx._scen_c(sc)
    nbytes = int(lib.mpc_solve_workspace_bytes(3, N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ctx.device)
    x0 = torch.as_tensor(hs.x).to(ctx.device)
    left = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    st = C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)
    call = lambda fn, *a: _ffi.check(ctx._ctx, fn(ctx._ctx, *a))

    def begin():
        call(lib.mpc_solve_begin, C.byref(scc), x0.data_ptr(), 1e-3, 50, ws.data_ptr(), N, nbytes, st)

    def rounds(k, bound):
        call(lib.mpc_solve_rounds, C.byref(scc), ws.data_ptr(), N, nbytes, k, bound, left.data_ptr(), st)
        return int(left.item())

    begin()
    assert rounds(1, N) == N                       # every problem asks for its first QP
    assert rounds(0, N) == N and rounds(3, 0) == N  # nothing launched: the count as it stands
    assert rounds(1, N // 2) == -1                 # half the problems left out of the round
    assert rounds(2, N) == -1                      # ... and the flag stays up
    begin()                                        # a new solve on the same workspace clears it
    n = N
    for _ in range(200):
        n = rounds(6, n)
        assert n >= 0
        if n == 0:
            break
    assert n == 0
    out = [torch.empty(N, mp.n, dtype=torch.float64, device=ctx.device), torch.empty(N, dtype=torch.float64, device=ctx.device),
           torch.empty(N, dtype=torch.int32, device=ctx.device), torch.empty(N, dtype=torch.int32, device=ctx.device)]
    result = lambda: call(lib.mpc_solve_result, N, ws.data_ptr(), N, nbytes, *(t.data_ptr() for t in out), None, st)
    result()
    x1, status1 = out[0].clone(), out[3].clone()
    assert rounds(6, N) == 0 and rounds(6, 0) == 0  # harmless after the end
    result()
    assert torch.equal(out[0], x1) and torch.equal(out[3], status1) and bool((status1 == 0).all())


def test_qp_step_rejects_an_index_outside_the_state_rows():
    """mpc_qp_step never addresses state outside [0, rows): such a request gets mode -100 and the valid requests
    of the same call are served as usual."""
    import ctypes as C

    from helpers import host_scenarios
    from mpc_amd import _ffi
    from mpc_amd.batch import MpcContext
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    rows = 4
    ctx = MpcContext(mp, 8, 0)
    _, hs = host_scenarios(mp, rows, motion="linear", cfg_id=24)
    sc, x = to_device(ctx, hs)
    r = ctx.eval_fd_nz(sc, x)
    n, m = mp.n, mp.m
    dev = ctx.device
    z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
    L, R, V = z(rows, (n + 1) * n // 2 + 1), z(rows, m + 2 * n), z(rows, n)
    lim = torch.as_tensor(np.tile(mp.v_lim, mp.horizon)).to(dev)
    index = torch.tensor([1, rows, -3, 2], dtype=torch.int32, device=dev)
    flags = torch.ones(4, dtype=torch.int32, device=dev)             # reset the factor, then solve
    lo, hi = (-lim - x[[1, 0, 0, 2]]).contiguous(), (lim - x[[1, 0, 0, 2]]).contiguous()
    s_out, r_out = z(4, n), z(4, m)
    mode = torch.zeros(4, dtype=torch.int32, device=dev)
    before = (L.clone(), R.clone(), V.clone())
    _ffi.check(ctx._ctx, ctx._lib.mpc_qp_step(ctx._ctx, 4, rows, index.data_ptr(), flags.data_ptr(), lo.data_ptr(), hi.data_ptr(),
                                              z(4, n).data_ptr(), L.data_ptr(), r.grad.data_ptr(), r.con.data_ptr(),
                                              r.con_jac_nz.data_ptr(), R.data_ptr(), V.data_ptr(), s_out.data_ptr(),
                                              r_out.data_ptr(), mode.data_ptr(), ctx._stream()))
    torch.cuda.synchronize()
    assert mode.cpu().tolist()[1:3] == [-100, -100] and mode[0].item() == 1 and mode[3].item() == 1
    for k in (0, 3):                                                  # rows that no request named are untouched
        assert torch.equal(L[k], before[0][k]) and torch.equal(R[k], before[1][k]) and torch.equal(V[k], before[2][k])
    assert float(L[1].abs().sum()) > 0 and float(s_out[0].abs().sum()) > 0 and float(s_out[1].abs().sum()) == 0


def test_qp_workspace_allocation_failure_is_reported_and_recoverable():
    """mpc_qp_step allocates its teams' workspace on first use (~0.5 GB at H = 20 for a large context).  With the
    device's memory taken, the call fails with the HIP error in the context's message, launches nothing (state rows
    untouched) and leaves the context usable: the same call succeeds once memory is free again."""
    from helpers import host_scenarios
    from mpc_amd import _ffi
    from mpc_amd.batch import MpcContext
    from mpc_amd.scenarios import to_device

    mp = MpcParams.reference_yaml_defaults(horizon=20)
    rows = 4
    warm = MpcContext(mp, 8, 0)                                        # code objects loaded, small workspace
    ctx = MpcContext(mp, 65536, 0)                                     # 512 teams: ~0.5 GB on first use
    _, hs = host_scenarios(mp, rows, motion="linear", cfg_id=24)
    n, m = mp.n, mp.m
    dev = ctx.device
    z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)

    def step(c):
        sc, x = to_device(c, hs)
        r = c.eval_fd_nz(sc, x)
        L, R, V = z(rows, (n + 1) * n // 2 + 1), z(rows, m + 2 * n), z(rows, n)
        lim = torch.as_tensor(np.tile(mp.v_lim, mp.horizon)).to(dev)
        index = torch.arange(rows, dtype=torch.int32, device=dev)
        flags = torch.ones(rows, dtype=torch.int32, device=dev)
        lo, hi = (-lim - x).contiguous(), (lim - x).contiguous()
        s_out, r_out = z(rows, n), z(rows, m)
        mode = torch.zeros(rows, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        return L, mode, s_out, lambda: c._lib.mpc_qp_step(
            c._ctx, rows, rows, index.data_ptr(), flags.data_ptr(), lo.data_ptr(), hi.data_ptr(), z(rows, n).data_ptr(),
            L.data_ptr(), r.grad.data_ptr(), r.con.data_ptr(), r.con_jac_nz.data_ptr(), R.data_ptr(), V.data_ptr(),
            s_out.data_ptr(), r_out.data_ptr(), mode.data_ptr(), c._stream()), (r, R, V, lo, hi, index, flags, r_out)

    Lw, mode_w, s_w, call_w, keep_w = step(warm)
    assert call_w() == 0
    torch.cuda.synchronize()
    assert mode_w.cpu().tolist() == [1] * rows
    L, mode, s_out, call, keep = step(ctx)
    hog = []                                                           # take the device's memory down to < 64 MB free
    for chunk in (8 << 30, 1 << 30, 64 << 20):
        try:
            while len(hog) < 4096:
                hog.append(torch.empty(chunk, dtype=torch.uint8, device=dev))
        except (torch.cuda.OutOfMemoryError, RuntimeError):
            pass
    rc = call()
    assert rc != 0
    with pytest.raises(_ffi.MpcError, match="memory|hipMalloc|alloc"):
        _ffi.check(ctx._ctx, rc)
    torch.cuda.synchronize()
    del hog
    torch.cuda.empty_cache()
    assert float(L.abs().sum()) == 0 and mode.cpu().tolist() == [0] * rows    # nothing ran
    assert call() == 0
    torch.cuda.synchronize()
    assert mode.cpu().tolist() == [1] * rows and torch.equal(s_out, s_w) and torch.equal(L, Lw)


def test_prepare_tick_rejects_an_in_place_warm_start_shift():
    """x0[i] = prev_solution[i + 6]: the same buffer for both would be read after it was overwritten."""
    import ctypes as C

    from mpc_amd import _ffi
    from mpc_amd.batch import MpcContext

    mp = MpcParams.reference_yaml_defaults(horizon=3)
    N = 4
    ctx = MpcContext(mp, N, 0)
    z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=ctx.device)
    q, ee, base, prev = z(N, 6), z(N, 6), z(N, 12), z(N, mp.n)
    has = torch.ones(N, dtype=torch.uint8, device=ctx.device)
    st, bt, jc, ui = z(N, 18), z(N, 3, 12), z(N, 36), z(N, 6)
    act = torch.zeros(N, dtype=torch.uint8, device=ctx.device)
    dyn = mp.base_dyn_array()
    rc = ctx._lib.mpc_prepare_tick(ctx._ctx, N, q.data_ptr(), None, ee.data_ptr(), base.data_ptr(),
                                   dyn.ctypes.data_as(C.POINTER(C.c_double)), prev.data_ptr(), has.data_ptr(), st.data_ptr(),
                                   bt.data_ptr(), jc.data_ptr(), ui.data_ptr(), prev.data_ptr(), act.data_ptr(), ctx._stream())
    assert rc == -1 and b"overlap" in ctx._lib.mpc_last_error(ctx._ctx)


def test_ordered_sum_enters_the_chain_at_every_term():
    """lsq_wave.hpp's ordered_sum_from (a computed jump into 128 back-to-back DPP instructions; every recurrence of the
    QP kernel and every sum of the body kernel runs through it) against the plain loop, bit for bit, for every entry
    point first = 0..128, with terms whose sum depends on the order of the additions."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    ctx = MpcContext(MpcParams.reference_yaml_defaults(horizon=1), 1, 0)
    rng = np.random.default_rng(7)
    reps = 3
    first = np.tile(np.arange(129, dtype=np.int32), reps)
    N = len(first)
    terms = rng.standard_normal((N, 128)) * np.exp(rng.uniform(-30, 30, (N, 128)))     # magnitudes over 26 decades
    t0 = rng.standard_normal(N)
    t0[:129] = 0.0
    want = t0.copy()
    for i in range(N):
        t = want[i]
        for j in range(first[i], 128):
            t = t + terms[i, j]
        want[i] = t
    dev = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt).to(ctx.device)
    got = ctx.ordered_sum_probe(dev(terms), dev(t0), dev(first, torch.int32)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # the order matters for these terms: a reversed chain gives other bits somewhere
    rev = np.array([np.sum(terms[i, first[i]:][::-1]) for i in range(200)])
    assert (rev != want[:200]).any()
