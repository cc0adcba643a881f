#!/bin/bash
# usage: tools/ab_many.sh <reps> <lib> [<lib> ...]   (paths of libmpc_amd variants; "cur" = the tree's own)
reps=$1; shift
for i in $(seq $reps); do
  for v in "$@"; do
    lib=$v; [ $v = cur ] && lib=mpc_amd/libmpc_amd.so
    MPC_AMD_LIB=$lib timeout -k 10 200 python -u tools/resident_time.py 65536 1 6 --no-devqp 2>&1 | grep resident | sed "s|^|$v |" | cut -c1-90
  done
done
