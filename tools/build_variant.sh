#!/bin/bash
# Build libmpc_amd from the sources of a git revision into ab_libs/<name>.so (for A/B runs with MPC_AMD_LIB).
# usage: tools/build_variant.sh <rev> <name> [extra hipcc flags]
set -e
rev=$1; name=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/ab_libs/${name}_src
rm -rf "$dst"; mkdir -p "$dst/csrc" "$dst/include"
git -C "$root" archive "$rev" mpc_amd/csrc include | tar -x -C "$dst" --strip-components=0
src=$dst/mpc_amd/csrc
objs=""
units="mpc_abi kernels_rollout kernels_aux kernels_prefix kernels_qp kernels_slsqp"
[ -f "$src/kernels_team.hip" ] && units="$units kernels_team"
for u in $units; do
  strict="-ffp-contract=off"; [ $u = mpc_abi ] && strict=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -I"$dst/include" -I"$src" $strict "$@" -c "$src/$u.hip" -o "$dst/$u.o" &
  objs="$objs $dst/$u.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/ab_libs/$name.so" $objs
echo "$root/ab_libs/$name.so"
