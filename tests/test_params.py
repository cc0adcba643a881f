"""CPU: config loader follows rospy.get_param precedence (YAML value, else the code default)."""
import numpy as np
import pytest

from mpc_amd.params import MpcParams

YAML_TREE = {
    "sim": True, "horizon": 1, "dt": 0.07,
    "weights": {"position": 20.0, "orientation": 3.0, "control": 0.1, "smooth": 0.3, "damping": 2.0},
    "active_weights": {"linear": 30.0, "angular": 30.0},
    "base_dynamics": {"use_dynamic_model": True, "mass": 65.0, "I_z": 8.0, "k_v": 25.0, "k_omega": 4.0,
                      "tau_linear": 0.45, "tau_angular": 0.30},
}


def test_yaml_values_win():
    p = MpcParams.from_param_tree(YAML_TREE)
    assert p == MpcParams.reference_yaml_defaults()
    assert (p.horizon, p.dt, p.dyn_dt) == (1, 0.07, 0.07)


def test_code_defaults_when_key_absent():
    p = MpcParams.from_param_tree({})
    # ur_mpc.py:29-30, ur20_dynamics.py:19 read the same key with different defaults
    assert (p.horizon, p.dt, p.dyn_dt) == (10, 0.1, 0.05)
    assert (p.w_position, p.w_linear, p.w_angular, p.w_damping) == (3.0, 50.0, 5.0, 0.5)
    assert p.use_dynamic_model is False and p.base_mass == 180.0


def test_validation_like_reference():
    with pytest.raises(ValueError, match="Horizon must be positive"):
        MpcParams(horizon=0)
    with pytest.raises(ValueError, match="Time step must be positive"):
        MpcParams(dt=0.0)


def test_decay_and_bounds_and_sizes():
    p = MpcParams.reference_yaml_defaults(horizon=20)
    assert (p.n, p.m) == (120, 252)
    d = p.decay_table()
    assert all(d[k] == 0.95 ** k for k in range(20))
    b = p.bounds()
    assert len(b) == 120 and b[7] == (-2.094395102393195, 2.094395102393195)
    c = p.to_c()
    assert c.horizon == 20 and c.dt_cost == 0.07 and c.q_hi[3] == 2 * np.pi
