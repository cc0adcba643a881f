"""ctypes binding of libmpc_amd.so (include/mpc_amd.h).  There is no CPU fallback: if the
library is missing or no HIP device is visible, the calls raise."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPC_AMD_LIB") or os.path.join(_HERE, "libmpc_amd.so")  # override: A/B builds

MPC_OK = 0
ALGO_AUTO, ALGO_ROLLOUT, ALGO_PREFIX = 0, 1, 2
MAX_HORIZON = 255
ABI_VERSION = 11
STATIC_SCRATCH_DOUBLES = 2 * 1024 * 1024

EXPORTS = (
    "mpc_abi_version", "mpc_ctx_create", "mpc_ctx_destroy", "mpc_set_params", "mpc_eval", "mpc_eval_fd",
    "mpc_jacobian", "mpc_fk", "mpc_predict_base_dynamic", "mpc_predict_base_dynamic_cmd", "mpc_prepare_tick", "mpc_qp_solve", "mpc_qp_step", "mpc_gather_scenarios", "mpc_jacobian_nz_count", "mpc_eval_fd_nz",
    "mpc_expand_jacobian", "mpc_last_error", "mpc_eval_fd_static",
    "mpc_sincos_probe", "mpc_ordered_sum_probe", "mpc_solve_workspace_bytes", "mpc_solve_begin", "mpc_solve_rounds", "mpc_solve_run", "mpc_solve_result",
)


class MpcParamsC(C.Structure):
    _fields_ = [
        ("horizon", C.c_int32), ("algo", C.c_int32),
        ("dt_cost", C.c_double), ("dt_con", C.c_double),
        ("w_position", C.c_double), ("w_orientation", C.c_double), ("w_control", C.c_double),
        ("w_smooth", C.c_double), ("w_damping", C.c_double), ("w_linear", C.c_double),
        ("w_angular", C.c_double),
        ("terminal_factor", C.c_double), ("fd_eps", C.c_double),
        ("q_lo", C.c_double * 6), ("q_hi", C.c_double * 6), ("v_lim", C.c_double * 6),
        ("dh_a", C.c_double * 6), ("dh_d", C.c_double * 6), ("dh_ca", C.c_double * 6),
        ("dh_sa", C.c_double * 6),
        ("base_c", C.c_double), ("base_s", C.c_double), ("tool", C.c_double * 3),
    ]


class MpcScenariosC(C.Structure):
    _fields_ = [
        ("count", C.c_int32), ("reserved", C.c_int32),
        ("state0", C.c_void_p), ("target", C.c_void_p), ("base_traj", C.c_void_p),
        ("Jc", C.c_void_p), ("u_prev", C.c_void_p), ("has_prev", C.c_void_p),
    ]


class MpcError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"libmpc_amd error {code}: {text}")
        self.code = code


_lib = None


def load():
    """Load the library once; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m mpc_amd.build` (hipcc, gfx950). "
            "mpc_amd has no CPU path.")
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
    L.mpc_abi_version.restype = ip
    L.mpc_abi_version.argtypes = []
    L.mpc_ctx_create.restype = ip
    L.mpc_ctx_create.argtypes = [ip, ip, ip, C.POINTER(vp)]
    L.mpc_ctx_destroy.restype = None
    L.mpc_ctx_destroy.argtypes = [vp]
    L.mpc_set_params.restype = ip
    L.mpc_set_params.argtypes = [vp, C.POINTER(MpcParamsC), dp, ip]
    L.mpc_eval.restype = ip
    L.mpc_eval.argtypes = [vp, C.POINTER(MpcScenariosC), vp, vp, vp, vp]
    L.mpc_eval_fd.restype = ip
    L.mpc_eval_fd.argtypes = [vp, C.POINTER(MpcScenariosC), vp, vp, vp, vp, vp, vp]
    L.mpc_eval_fd_static.restype = ip
    L.mpc_eval_fd_static.argtypes = [vp, C.POINTER(MpcScenariosC), vp, vp, vp, vp, vp, vp]
    L.mpc_jacobian.restype = ip
    L.mpc_jacobian.argtypes = [vp, ip, vp, vp, vp]
    L.mpc_fk.restype = ip
    L.mpc_sincos_probe.restype = ip
    L.mpc_sincos_probe.argtypes = [vp, ip, vp, vp, vp, vp]
    L.mpc_ordered_sum_probe.restype = ip
    L.mpc_ordered_sum_probe.argtypes = [vp, ip, vp, vp, vp, vp, vp]
    L.mpc_fk.argtypes = [vp, ip, vp, vp, vp]
    L.mpc_predict_base_dynamic.restype = ip
    L.mpc_predict_base_dynamic.argtypes = [vp, ip, vp, dp, vp, vp]
    L.mpc_predict_base_dynamic_cmd.restype = ip
    L.mpc_predict_base_dynamic_cmd.argtypes = [vp, ip, vp, vp, dp, vp, vp]
    L.mpc_gather_scenarios.restype = ip
    L.mpc_gather_scenarios.argtypes = [vp, C.POINTER(MpcScenariosC), vp, ip, C.POINTER(MpcScenariosC), vp]
    L.mpc_prepare_tick.restype = ip
    L.mpc_prepare_tick.argtypes = [vp, ip, vp, vp, vp, vp, dp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_qp_solve.restype = ip
    L.mpc_qp_solve.argtypes = [vp, ip, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_qp_step.restype = ip
    L.mpc_qp_step.argtypes = [vp, ip, ip] + [vp] * 15
    L.mpc_jacobian_nz_count.restype = ip
    L.mpc_jacobian_nz_count.argtypes = [ip]
    L.mpc_eval_fd_nz.restype = ip
    L.mpc_eval_fd_nz.argtypes = [vp, C.POINTER(MpcScenariosC), vp, vp, vp, vp, vp, vp]
    L.mpc_expand_jacobian.restype = ip
    L.mpc_expand_jacobian.argtypes = [vp, ip, vp, vp, vp]
    L.mpc_solve_workspace_bytes.restype = C.c_size_t
    L.mpc_solve_workspace_bytes.argtypes = [ip, ip]
    L.mpc_solve_begin.restype = ip
    L.mpc_solve_begin.argtypes = [vp, C.POINTER(MpcScenariosC), vp, C.c_double, ip, vp, ip, C.c_size_t, vp]
    L.mpc_solve_rounds.restype = ip
    L.mpc_solve_rounds.argtypes = [vp, C.POINTER(MpcScenariosC), vp, ip, C.c_size_t, ip, ip, vp, vp]
    L.mpc_solve_run.restype = ip
    L.mpc_solve_run.argtypes = [vp, C.POINTER(MpcScenariosC), vp, ip, C.c_size_t, ip, vp, vp]
    L.mpc_solve_result.restype = ip
    L.mpc_solve_result.argtypes = [vp, ip, vp, ip, C.c_size_t, vp, vp, vp, vp, vp, vp]
    L.mpc_last_error.restype = C.c_char_p
    L.mpc_last_error.argtypes = [vp]
    if L.mpc_abi_version() != ABI_VERSION:
        raise ImportError(f"libmpc_amd ABI {L.mpc_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(ctx, rc: int):
    if rc != MPC_OK:
        raise MpcError(rc, load().mpc_last_error(ctx).decode("utf-8", "replace"))
