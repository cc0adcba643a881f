#!/bin/bash
# usage: tools/pmc_qp.sh <outdir> [K]   -- counter passes over mpc_qp_solve (tools/qp_time.py), one run per pass
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
out=$1; shift
mkdir -p $out
i=0
for set in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_WAVE32_LDS SQ_ACTIVE_INST_FLAT" \
 "FETCH_SIZE" \
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 $R/tools/qp_time.py "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "qp_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
