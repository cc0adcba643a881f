#!/bin/bash
# per-round durations of the QP kernel in one resident solve: tools/round_times.sh <lib> [<lib> ...]   ("cur" = the tree's own)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/$v; [ $v = cur ] && lib=$GRAFT_REPO_ROOT/mpc_amd/libmpc_amd.so
  name=$(basename $v .so)
  export MPC_AMD_LIB=$lib
  rm -rf /tmp/rt_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rt_$name -o p -- python3 -u $GRAFT_REPO_ROOT/tools/resident_time.py 65536 1 6 --no-devqp > /tmp/rt_$name.log 2>&1 || { tail -5 /tmp/rt_$name.log; exit 1; }
  f=$(find /tmp/rt_$name -name "*kernel_trace.csv" | head -1)
  echo "== $v"; grep resident /tmp/rt_$name.log | cut -c1-60
  python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
qp = [(s, e) for s, e, n in k if 'qp_step_kernel' in n]
last = qp[-174:]                      # the last (timed) solve
print('QP rounds (ms):', ' '.join('%.2f' % ((e - s) / 1e6) for s, e in last[:40]))
print('QP total %.1f ms, span %.1f ms' % (sum(e - s for s, e in last) / 1e6, (last[-1][1] - last[0][0]) / 1e6))
P
done
