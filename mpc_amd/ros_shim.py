"""ROS-side drop-in: ``from mpc_amd.ros_shim import URMPC`` where the action server does
``from ur20_mpc_controller.models.ur_mpc import URMPC`` (mpc_action_server.py:7) and then constructs it
with no arguments (mpc_action_server.py:30).

The reference's constructor reads its settings key by key with ``rospy.get_param('~mpc_controller/<key>',
<code default>)`` (ur_mpc.py:29-81, ur20_dynamics.py:19); here the whole private namespace is fetched once and
resolved with the same precedence by ``MpcParams.from_param_tree``.  Nothing else of ROS is touched: the
action server, MoveIt, the base observer and Gazebo stay as they are.
"""
from __future__ import annotations

from .params import MpcParams
from .ur_mpc import GpuURMPC


def params_from_ros() -> MpcParams:
    import rospy  # only importable on a ROS host; tests install a stub

    return MpcParams.from_param_tree(rospy.get_param("~mpc_controller", {}))


def URMPC(device: int = 0) -> GpuURMPC:
    """No-argument factory with the reference class's name."""
    return GpuURMPC(params_from_ros(), device=device)
