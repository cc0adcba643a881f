"""CPU: host-side glue of the drop-in layer that needs no GPU."""
import os

import numpy as np
import pytest

from mpc_amd.params import MpcParams
from mpc_amd.scenarios import dls_guess, make_scenarios, world_to_base
from mpc_amd.ur_mpc import KinematicBasePredictor, _pack_base, _unpack_base_traj
from helpers import host_scenarios, params_for
from oracle.numpy_ref import NumpyRefMPC, RefParams


def test_kinematic_predictor_matches_reference_golden(golden_dir):
    path = os.path.join(golden_dir, "base_motion.npz")
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path, allow_pickle=False)
    H = int(g["H"])
    pred = KinematicBasePredictor()
    for t in range(g["base"].shape[0]):
        b = g["base"][t]
        bs = {"position": b[0:3], "orientation": b[3:6], "linear_velocity": b[6:9], "angular_velocity": b[9:12]}
        got = np.stack([_pack_base(e) for e in pred.predict(bs, H, 0.07)])
        np.testing.assert_array_equal(got, g["kinematic"][t])


def test_pack_roundtrip():
    rng = np.random.default_rng(0)
    arr = rng.standard_normal((5, 12))
    back = np.stack([_pack_base(e) for e in _unpack_base_traj(arr)])
    np.testing.assert_array_equal(back, arr)


def test_batched_dls_guess_matches_reference_formula():
    mp = params_for(3)
    co, hs = host_scenarios(mp, 300, motion="angular", cfg_id=3)
    ref = NumpyRefMPC(RefParams(horizon=3))
    for i in range(300):
        b = hs.base[i]
        bs = {"position": b[0:3], "orientation": b[3:6], "linear_velocity": b[6:9], "angular_velocity": b[9:12]}
        ee = {"position": hs.state0[i, 12:15], "orientation": hs.state0[i, 15:18]}
        want = ref.dls_guess(hs.Jc[i].reshape(6, 6), bs, ee)
        np.testing.assert_array_equal(hs.u_init[i], want)      # same operation order as ur_mpc.py:173-175: same bits


def test_scenarios_are_shard_stable_and_in_bounds():
    mp = params_for(20)
    co, full = host_scenarios(mp, 40, motion="linear", cfg_id=2)
    _, part = host_scenarios(mp, 15, motion="linear", cfg_id=2, first=25)
    for name in ("x", "state0", "target", "base_traj", "Jc"):
        np.testing.assert_array_equal(getattr(full, name)[25:], getattr(part, name))
    lim = np.tile(np.asarray(mp.v_lim), 20)
    assert np.all(np.abs(full.x) <= lim)
    # platform family of the reference's simulator: 0.1 Hz, amplitude 0.05-0.30 m on x
    w = 2 * np.pi * 0.1
    assert np.all(np.abs(full.base[:, 6]) <= 0.30 * w + 1e-12) and np.all(full.base[:, [1, 2, 7, 8]] == 0)


def test_sweep_is_the_product_of_the_lattice_and_the_goal_table():
    """BASELINE configs[4]: global row i = (amplitude ia, phase ip, goal j) with i = (ia*32 + ip)*1024 + j, whatever
    shard it is made in; ranges of SURVEY.md 8(d) (simulate_platform_motion.py:17,23,42-59); arm at the SRDF pose."""
    from mpc_amd.scenarios import (Q_ANGULAR_READY, SWEEP_ROWS, make_sweep, sweep_goal_table, sweep_index, sweep_lattice)
    from oracle.oracle import OracleKinematics
    from helpers import oracle_for

    mp = params_for(20)
    kin = OracleKinematics(oracle_for(mp))
    assert SWEEP_ROWS == 1 << 20
    first = 5 * 32 * 1024 + 7 * 1024 - 9                      # straddles the cells (5, 6) and (5, 7)
    full = make_sweep(mp, kin, first, 40, motion="angular")
    part = make_sweep(mp, kin, first + 25, 15, motion="angular")
    for name in ("x", "state0", "target", "base_traj", "Jc", "base", "active"):
        np.testing.assert_array_equal(getattr(full, name)[25:], getattr(part, name))
    ia, ip, goal = sweep_index(first + np.arange(40))
    assert (ia == 5).all() and list(ip[:9]) == [6] * 9 and list(ip[9:]) == [7] * 31 and goal[9] == 0 and goal[8] == 1023
    A, phi = sweep_lattice("angular")
    assert np.radians(5.0) < A[0] < A[-1] < np.radians(30.0) and 0 < phi[0] < phi[-1] < 2 * np.pi and len(A) == len(phi) == 32
    w = 2 * np.pi * 0.1
    np.testing.assert_array_equal(full.base[:, 5], A[ia] * np.sin(phi[ip]))
    np.testing.assert_array_equal(full.base[:, 11], A[ia] * w * np.cos(phi[ip]))
    np.testing.assert_array_equal(full.q0, np.tile(Q_ANGULAR_READY, (40, 1)))
    # the same goal (offset from the tip pose) in two different lattice cells
    other = make_sweep(mp, kin, 17 * 32 * 1024 + 3 * 1024 + int(goal[0]), 1, motion="angular")
    tab = sweep_goal_table(20)
    np.testing.assert_allclose(other.target[0] - other.state0[0, 12:18], tab[goal[0], :6], rtol=0, atol=1e-15)
    np.testing.assert_allclose(full.target[0] - full.state0[0, 12:18], tab[goal[0], :6], rtol=0, atol=1e-15)
    lin = make_sweep(mp, kin, SWEEP_ROWS - 3, 3, motion="linear")
    Al, _ = sweep_lattice("linear")
    assert 0.05 < Al[0] < Al[-1] < 0.30 and np.all(lin.base[:, [1, 2, 3, 4, 5, 7, 8]] == 0)
    assert np.all(np.abs(lin.x) <= np.tile(np.asarray(mp.v_lim), 20))
    with pytest.raises(ValueError):
        make_sweep(mp, kin, SWEEP_ROWS - 3, 4)


def test_batched_kinematic_predictor_equals_per_controller_predictor():
    """Rows of the batched predictor == one KinematicBasePredictor per controller, bit for bit, over ticks on
    which only some controllers predict (the others took the early exit and keep their filter state)."""
    from mpc_amd.solver import BatchedKinematicPredictor
    from mpc_amd.ur_mpc import KinematicBasePredictor

    rng = np.random.default_rng(3)
    N, H, dt = 9, 5, 0.07
    batched = BatchedKinematicPredictor(N)
    single = [KinematicBasePredictor() for _ in range(N)]
    for tick in range(6):
        base = rng.normal(0, 0.4, (N, 12))
        rows = np.nonzero(rng.random(N) < 0.7)[0] if tick else np.arange(0, N, 2)
        got = batched.predict(rows, base[rows], H, dt)
        for j, r in enumerate(rows):
            b = base[r]
            want = single[r].predict({"position": b[0:3], "orientation": b[3:6], "linear_velocity": b[6:9],
                                      "angular_velocity": b[9:12]}, H, dt)
            for k in range(H):
                row = np.concatenate([want[k]["position"], want[k]["orientation"], want[k]["linear_velocity"],
                                      want[k]["angular_velocity"]])
                np.testing.assert_array_equal(got[j, k], row)


def test_device_qp_driver_lane_choice():
    """One lane per ~min_per_lane problems, never more than there are, always enough for the parts to fit."""
    from mpc_amd.native_solver import active_lanes

    assert active_lanes(9, 4, 64, 32) == 1
    assert active_lanes(150, 4, 64, 32) == 4
    assert active_lanes(100, 4, 64, 32) == 3
    assert active_lanes(4095, 4, 2048, 2048) == 2          # one lane would not hold it
    assert active_lanes(65536, 4, 16384, 2048) == 4
    assert active_lanes(3000, 4, 16384, 2048) == 1
    for count in range(1, 700, 7):
        k = active_lanes(count, 4, 175, 50)
        assert 1 <= k <= 4 and (count + k - 1) // k <= 175
