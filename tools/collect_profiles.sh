#!/bin/bash
# usage: bash tools/collect_profiles.sh gpurun_out/<dir> profiles/rNN   -- copies the summaries worth judging, builds the JSON
set -e
src=$1; dst=$2
mkdir -p $dst
cp $src/bench_line*.json $src/bench_sweep*.json $src/pmc_summary_*.txt $src/qp_solve_4096.txt $src/solve_cfg3_65536_resident.txt $dst/
for k in kt_h20:kernel_stats_bench_only_main kt_h100:kernel_stats_bench_only_main_H100_N16384 kt_qp:kernel_stats_qp_solve_4096 kt_solve:kernel_stats_solve_cfg3_65536_resident; do
  f=$(find $src/${k%%:*} -name "*kernel_stats.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2-)    # the newest: merges pile up
  [ -n "$f" ] && cp $f $dst/${k##*:}.csv
done
t=$(find $src/kt_solve -name "*kernel_trace.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2-)
[ -n "$t" ] && python3 - "$t" $dst/kernel_trace_solve_cfg3_65536_resident.csv <<'P'
import csv, sys          # the trace, reduced to what the share needs: name (shortened), start and end of every dispatch
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
for r in rows:
    w.writerow([r["Kernel_Name"][:70], r["Start_Timestamp"], r["End_Timestamp"]])
P
python3 tools/profiles_json.py $dst --qp $dst/pmc_summary_qp_kernel_4096.txt:20:4096:$dst/kernel_stats_solve_cfg3_65536_resident.csv:$dst/kernel_trace_solve_cfg3_65536_resident.csv \
  $dst/pmc_summary_prefix_H20_N65536.txt:jac:20:65536 \
  $dst/pmc_summary_prefix_H20_N65536_gradient_only.txt:nojac:20:65536 \
  $dst/pmc_summary_prefix_H20_N65536_structured.txt:nz:20:65536 \
  $dst/pmc_summary_H100_N16384.txt:rollout:100:16384:prefix_kernel:16 \
  $dst/pmc_summary_H100_N16384.txt:expand:100:16384:expand_kernel:16
