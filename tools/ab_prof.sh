#!/bin/bash
# per-kernel totals of one resident solve for each library variant: tools/ab_prof.sh <lib> [<lib> ...]
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/$v; [ $v = cur ] && lib=$GRAFT_REPO_ROOT/mpc_amd/libmpc_amd.so
  name=$(basename $v .so)
  export MPC_AMD_LIB=$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 -u $GRAFT_REPO_ROOT/tools/resident_time.py 65536 1 6 --no-devqp > /tmp/prof_$name.log 2>&1 || { tail -5 /tmp/prof_$name.log; exit 1; }
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep resident /tmp/prof_$name.log | cut -c1-60
  if [ -z "$f" ]; then tail -5 /tmp/prof_$name.log; find /tmp/prof_$name | head; else python3 -c "
import csv,sys
for i,r in enumerate(csv.DictReader(open('$f'))):
    if i<7: print('%-70s %6s %12.3f ms' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6))
"; fi
done
