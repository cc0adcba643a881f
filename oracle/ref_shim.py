"""TEST INFRASTRUCTURE — loads the *real* reference (``/root/reference``) in this
container under ROS stubs, to pin the oracle and to generate golden vectors.

Nothing here copies reference source; it only puts the reference package on
``sys.path`` and provides the three ROS modules its imports need (SURVEY.md
Appendix B).  The reference does not exist on the GPU box, so every user of
this module must tolerate ``reference_available() == False``.

The stubbed ``MoveGroupCommander.get_jacobian_matrix`` delegates to this
repo's own UR20 Jacobian (``oracle.numpy_ref.ur20_jacobian``): MoveIt and the
UR20 URDF are not in the reference tree, so the Jacobian boundary is
"parity unpinned" and everything *else* is compared exactly.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
_PKG_PARENT = os.path.join(REFERENCE_ROOT, "ur20_mpc_controller")
_YAML = os.path.join(_PKG_PARENT, "config", "controller_params.yaml")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_PKG_PARENT, "ur20_mpc_controller", "models", "ur_mpc.py"))


class _ParamServer:
    """rospy.get_param backed by controller_params.yaml + overrides."""

    def __init__(self):
        self.tree = {}
        self.overrides = {}

    def load_yaml(self, path=_YAML):
        import yaml

        with open(path) as fh:
            self.tree = yaml.safe_load(fh) or {}

    def get(self, name, default=None):
        key = name.lstrip("~").strip("/")
        if key in self.overrides:
            return self.overrides[key]
        node = self.tree
        for part in key.split("/"):
            if isinstance(node, dict) and part in node:
                node = node[part]
            else:
                return default
        return node


PARAMS = _ParamServer()
_JAC_PROVIDER = [None]
JAC_CALLS = [0]


def _install_stubs():
    if "rospy" in sys.modules and getattr(sys.modules["rospy"], "_mpc_amd_stub", False):
        return
    rospy = types.ModuleType("rospy")
    rospy._mpc_amd_stub = True
    rospy.get_param = lambda name, default=None: PARAMS.get(name, default)
    for lvl in ("logdebug", "loginfo", "logwarn", "logerr"):
        setattr(rospy, lvl, lambda *a, **k: None)
    rospy.init_node = lambda *a, **k: None

    class ROSInterruptException(Exception):
        pass

    rospy.ROSInterruptException = ROSInterruptException
    sys.modules["rospy"] = rospy

    mc = types.ModuleType("moveit_commander")
    mc.roscpp_initialize = lambda argv: None

    class RobotCommander:
        pass

    class MoveGroupCommander:
        def __init__(self, name):
            self.name = name

        def get_jacobian_matrix(self, q):
            JAC_CALLS[0] += 1
            J = _JAC_PROVIDER[0](np.asarray(q, dtype=float))
            return [list(map(float, row)) for row in np.asarray(J)]  # nested lists, like MoveIt

    mc.RobotCommander = RobotCommander
    mc.MoveGroupCommander = MoveGroupCommander
    sys.modules["moveit_commander"] = mc

    tf = types.ModuleType("tf")
    tft = types.ModuleType("tf.transformations")

    def euler_from_quaternion(qxyzw):
        import math

        x, y, z, w = qxyzw
        roll = math.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
        pitch = math.asin(max(-1.0, min(1.0, 2 * (w * y - z * x))))
        yaw = math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
        return (roll, pitch, yaw)

    tft.euler_from_quaternion = euler_from_quaternion
    tf.transformations = tft
    sys.modules["tf"] = tf
    sys.modules["tf.transformations"] = tft


def load_reference(overrides=None, jacobian=None, use_yaml=True):
    """Return a freshly constructed reference ``URMPC`` (real code, stubbed ROS).

    overrides: {'mpc_controller/horizon': 20, ...} applied on top of the YAML.
    jacobian:  callable q(6,) -> (6,6); defaults to the oracle's UR20 Jacobian.
    """
    if not reference_available():
        raise RuntimeError("reference tree not present")
    _install_stubs()
    if use_yaml:
        PARAMS.load_yaml()
    else:
        PARAMS.tree = {}
    PARAMS.overrides = dict(overrides or {})
    if jacobian is None:
        from oracle.numpy_ref import ur20_jacobian as jacobian
    _JAC_PROVIDER[0] = jacobian
    if _PKG_PARENT not in sys.path:
        sys.path.insert(0, _PKG_PARENT)
    from ur20_mpc_controller.models.ur_mpc import URMPC  # the reference itself

    return URMPC()
