#!/bin/bash
# usage: mkvar.sh name unit "defines"
set -e
R=/root/repo; name=$1; unit=$2; defs=$3
mkdir -p $R/ab_libs/v_$name
for u in mpc_abi kernels_rollout kernels_aux kernels_prefix kernels_qp kernels_slsqp; do
  if [ $u != $unit ]; then cp $R/mpc_amd/build/$u.o $R/ab_libs/v_$name/$u.o; fi
done
extra="-ffp-contract=off"; [ $unit = mpc_abi ] && extra=""
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$R/include -I$R/mpc_amd/csrc $defs $extra -c $R/mpc_amd/csrc/$unit.hip -o $R/ab_libs/v_$name/$unit.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_libs/lib_$name.so $R/ab_libs/v_$name/*.o
echo built $name
