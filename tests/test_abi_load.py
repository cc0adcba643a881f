"""CPU: the C-ABI library loads and exports every symbol include/mpc_amd.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from mpc_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_ffi.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_ffi.LIB_PATH):
        pytest.fail("libmpc_amd.so not built: run `python -m mpc_amd.build`")
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert _ffi.load().mpc_abi_version() == _ffi.ABI_VERSION


def test_slsqp_library_exports_every_declared_symbol():
    """include/mpc_slsqp.h against libmpc_slsqp.so (the native host driver of the batched solves)."""
    from mpc_amd import native_solver

    text = open(os.path.join(ROOT, "include", "mpc_slsqp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mpc_slsqp_[a-z_]+)\s*\(", text)))
    assert len(declared) >= 12
    if not os.path.exists(native_solver.LIB_PATH):
        pytest.fail("libmpc_slsqp.so not built: run `python -m mpc_amd.build`")
    lib = ctypes.CDLL(native_solver.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert native_solver.load().mpc_slsqp_abi_version() == native_solver.ABI_VERSION


def test_struct_sizes_match_header_layout():
    # 2 int32 + 11 scalars + 3*6 limits + 4*6 DH + 2 + 3 doubles
    assert ctypes.sizeof(_ffi.MpcParamsC) == 8 + 8 * (11 + 18 + 24 + 2 + 3)
    assert ctypes.sizeof(_ffi.MpcScenariosC) == 8 + 6 * 8


def test_no_cpu_fallback_without_gpu():
    """On a box without a HIP device context creation must fail loudly, never compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _ffi.load()
    ctx = ctypes.c_void_p()
    rc = lib.mpc_ctx_create(0, 20, 16, ctypes.byref(ctx))
    assert rc < 0 and not ctx.value
    assert b"no HIP device" in lib.mpc_last_error(None) or rc == -2


def test_argument_validation_messages():
    lib = _ffi.load()
    ctx = ctypes.c_void_p()
    assert lib.mpc_ctx_create(0, 0, 16, ctypes.byref(ctx)) == -1
    assert b"Horizon must be positive" in lib.mpc_last_error(None)   # ur_mpc.py:33-34
    assert lib.mpc_ctx_create(0, 10_000, 16, ctypes.byref(ctx)) == -3
    assert lib.mpc_ctx_create(0, 20, 16, None) == -1


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libmpc_oracle" not in src, f
