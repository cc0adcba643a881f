"""The ROS drop-in shim with a stubbed rospy: there is no ROS here or on the GPU box."""
import sys
import types

import numpy as np
import pytest

# the mpc_controller namespace of the reference's controller_params.yaml (values: config/controller_params.yaml:3-49)
YAML_TREE = {
    "sim": True, "horizon": 1, "dt": 0.07,
    "weights": {"position": 20.0, "orientation": 3.0, "control": 0.1, "smooth": 0.3, "damping": 2.0},
    "active_weights": {"linear": 30.0, "angular": 30.0},
    "arm_base_offset": {"x": 0.0, "y": 0.1, "z": 0.5, "yaw": 0.0},
    "base_dynamics": {"use_dynamic_model": True, "mass": 65.0, "I_z": 8.0, "wheel_radius": 0.16, "half_wheelbase": 0.638,
                      "k_v": 25.0, "k_omega": 4.0, "tau_linear": 0.45, "tau_angular": 0.30, "use_caster": False,
                      "tau_caster": 0.25},
}


@pytest.fixture
def fake_rospy(monkeypatch):
    calls = []
    mod = types.ModuleType("rospy")
    mod.tree = YAML_TREE

    def get_param(name, default=None):
        calls.append(name)
        assert name == "~mpc_controller"
        return mod.tree if mod.tree is not None else default

    mod.get_param = get_param
    mod.calls = calls
    monkeypatch.setitem(sys.modules, "rospy", mod)
    return mod


def test_params_come_from_the_private_namespace(fake_rospy):
    from mpc_amd.params import MpcParams
    from mpc_amd.ros_shim import params_from_ros

    assert params_from_ros() == MpcParams.reference_yaml_defaults()
    fake_rospy.tree = None                                   # nothing on the parameter server: code defaults
    p = params_from_ros()
    assert (p.horizon, p.dt, p.dyn_dt, p.use_dynamic_model) == (10, 0.1, 0.05, False)
    assert fake_rospy.calls == ["~mpc_controller", "~mpc_controller"]


def test_no_argument_constructor_without_gpu_fails_loudly(fake_rospy):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu test below")
    from mpc_amd import _ffi
    from mpc_amd.ros_shim import URMPC

    with pytest.raises((_ffi.MpcError, ImportError, RuntimeError)):
        URMPC()


@pytest.mark.gpu
def test_shim_controller_equals_explicit_construction(fake_rospy):
    from helpers import host_scenarios
    from mpc_amd.params import MpcParams
    from mpc_amd.ros_shim import URMPC
    from mpc_amd.ur_mpc import GpuURMPC

    a = URMPC()                                              # as mpc_action_server.py:30 does
    b = GpuURMPC(MpcParams.reference_yaml_defaults())
    assert a.dt == 0.07 and a.horizon == 1                   # the server reads .dt (mpc_action_server.py:54)
    _, hs = host_scenarios(b.params, 4, motion="linear", cfg_id=17)
    for i in range(4):
        args = ({"position": hs.q0[i], "velocity": np.zeros(6)},
                {"position": hs.state0[i, 12:15], "orientation": hs.state0[i, 15:18]},
                {"position": hs.target[i, :3], "orientation": hs.target[i, 3:]},
                {"position": hs.base[i, 0:3], "orientation": hs.base[i, 3:6], "linear_velocity": hs.base[i, 6:9],
                 "angular_velocity": hs.base[i, 9:12]})
        np.testing.assert_array_equal(a.compute_control(*args), b.compute_control(*args))
