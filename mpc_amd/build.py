"""Builds libmpc_amd.so (gfx950) in-tree with hipcc.  `python -m mpc_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libmpc_amd.so")
SLSQP_LIB_PATH = os.path.join(_HERE, "libmpc_slsqp.so")     # host-only: the native batched SLSQP driver
# the same with lsq's free step in its checking form (-DMPC_QP_FREE_STEP=2: the long form runs all the same and the
# library counts predictions that did not hold); tests/test_native_slsqp.py loads it through MPC_SLSQP_LIB
SLSQP_CHECK_LIB_PATH = os.path.join(_HERE, "libmpc_slsqp_check.so")
_SLSQP_UNITS = ["slsqp_core.c", "slsqp_batch.c"]
# one rounding per operation: with it the C core reproduces scipy's compiled core bit for bit
_SLSQP_FLAGS = ["-O3", "-ffp-contract=off", "-fPIC", "-fopenmp", "-Wall", "-std=gnu11"]

# translation unit -> extra flags.  "strict" units keep one rounding per operation.
_STRICT = ["-ffp-contract=off"]
_UNITS = {
    "mpc_abi.hip": [],
    "kernels_rollout.hip": _STRICT,
    "kernels_aux.hip": _STRICT,
    "kernels_prefix.hip": _STRICT,  # fused multiply-adds are written out with fma() there
    "kernels_qp.hip": _STRICT,      # bit-identical to the host SLSQP core
    "kernels_slsqp.hip": _STRICT,   # Kraft's body on the device: the host core's text, one rounding per operation
    "kernels_team.hip": _STRICT,    # whole solves per workgroup: the device code of the three units above, once more
}
_COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
           "-I" + os.path.join(_ROOT, "include"), "-I" + CSRC]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmpc_amd.so cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc", ".h"))]
    headers.append(os.path.join(_ROOT, "include", "mpc_amd.h"))
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, relink = [], force
    for unit, extra in _UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + _COMMON + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
    if relink or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_slsqp(force, verbose)
    return LIB_PATH


def build_slsqp(force: bool = False, verbose: bool = False) -> str:
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: libmpc_slsqp.so cannot be built")
    srcs = [os.path.join(CSRC, u) for u in _SLSQP_UNITS]
    deps = srcs + [os.path.join(CSRC, "slsqp_core.h"), os.path.join(CSRC, "slsqp_body.inc"),
                   os.path.join(_ROOT, "include", "mpc_slsqp.h")]
    spmd = [os.path.join(CSRC, "lsq_spmd_host.cpp"), os.path.join(CSRC, "lsq_spmd.hpp"), os.path.join(CSRC, "lsq_wave.hpp")]
    if force or _stale(SLSQP_LIB_PATH, deps + spmd) or _stale(SLSQP_CHECK_LIB_PATH, deps + spmd):
        objdir = os.path.join(_HERE, "build")
        os.makedirs(objdir, exist_ok=True)
        inc = ["-I" + os.path.join(_ROOT, "include"), "-I" + CSRC]
        spmd_obj = os.path.join(objdir, "lsq_spmd_host.o")
        gxx = shutil.which("g++") or gcc
        cmds = [[gxx, "-O2", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall"] + inc + ["-c", spmd[0], "-o", spmd_obj]]
        c_objs = []
        for src in srcs:
            obj = os.path.join(objdir, os.path.basename(src)[:-2] + ".o")
            c_objs.append(obj)
            cmds.append([gcc] + _SLSQP_FLAGS + inc + ["-c", src, "-o", obj])
        cmds.append([gxx, "-shared", "-fopenmp", "-o", SLSQP_LIB_PATH] + c_objs + [spmd_obj, "-lm"])
        check_obj = os.path.join(objdir, "lsq_spmd_host_check.o")
        cmds.append([gxx, "-O2", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-DMPC_QP_FREE_STEP=2"] + inc +
                    ["-c", spmd[0], "-o", check_obj])
        cmds.append([gxx, "-shared", "-fopenmp", "-o", SLSQP_CHECK_LIB_PATH] + c_objs + [check_obj, "-lm"])
        for cmd in cmds:
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return SLSQP_LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
