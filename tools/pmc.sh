#!/bin/bash
# usage: tools/pmc.sh <outdir> <prog> <prog args...>   -- runs the counter passes one by one (one rocprofv3 run per pass: the
# hardware cannot collect more than a handful of counters per block at once) and prints per-kernel means.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
out=$1; shift
prog=$1; shift
mkdir -p $out
i=0
for set in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
 "FETCH_SIZE" \
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VMEM"
do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 $R/tools/$prog "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        m = re.search(r"(prefix_kernel<[^>]*>|rollout_kernel<[^>]*>|expand_kernel|qp_step_kernel<[^>]*>|qp_kernel<[^>]*>|solve_body_kernel)", name)
        if m:
            agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kernel, counters in agg.items():
    print(f"== {kernel}")
    for k, v in counters.items():
        print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
