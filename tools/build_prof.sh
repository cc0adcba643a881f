#!/bin/bash
# Development only: libmpc_amd with the QP kernels' per-phase cycle counters (-DMPC_QP_PROFILE) -> ab_libs/libmpc_amd_prof.so
# (git-ignored; use it with MPC_AMD_LIB=ab_libs/libmpc_amd_prof.so python tools/qp_phases.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ab_libs/obj
C="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$R/include -I$R/mpc_amd/csrc -DMPC_QP_PROFILE $MPC_EXTRA_DEFINES"
for u in mpc_abi kernels_rollout kernels_aux kernels_prefix kernels_qp kernels_slsqp; do
  extra="-ffp-contract=off"; [ $u = mpc_abi ] && extra=""
  if [ $u = kernels_qp ] || [ ! -f $R/ab_libs/obj/$u.o ] || [ $R/mpc_amd/csrc/$u.hip -nt $R/ab_libs/obj/$u.o ]; then
    hipcc $C $extra -c $R/mpc_amd/csrc/$u.hip -o $R/ab_libs/obj/$u.o
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_libs/libmpc_amd_prof.so $R/ab_libs/obj/*.o
echo $R/ab_libs/libmpc_amd_prof.so
