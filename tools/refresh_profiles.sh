#!/bin/bash
# Runs on the GPU box: the bench line, the kernel-trace summaries of the same commands and the PMC passes.
# usage: bash tools/refresh_profiles.sh <outdir under gpurun_out>.  Stops at the first failing step.
# Afterwards, here: bash tools/collect_profiles.sh gpurun_out/<outdir> profiles/rNN
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python3 -u bench.py > $out/bench_line.json
echo "bench done"
timeout -k 10 300 python3 -u bench.py --workload sweep --motion angular --no-cpu-baseline > $out/bench_sweep_angular_131072.json
echo "bench (configs[4] shard) done"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_h20 -- python3 $R/bench.py --only-main > $out/bench_line_only_main.json 2> $out/kt_h20.log
echo "kernel trace H=20 done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_h100 -- python3 $R/bench.py --only-main --horizon 100 --scenarios 16384 --steps 10 > $out/bench_line_H100_only_main.json 2> $out/kt_h100.log
echo "kernel trace H=100 done"
cd $R
timeout -k 10 200 python3 -u bench.py --horizon 100 --scenarios 16384 --steps 10 --no-cpu-baseline --solve-sample 0 > $out/bench_line_H100.json
echo "bench H=100 done"
bash tools/pmc.sh $out/pmc prof.py 20 65536 0 jac > $out/pmc_summary_prefix_H20_N65536.txt
echo "pmc (dense dc/dx) done"
bash tools/pmc.sh $out/pmc_nojac prof.py 20 65536 0 nojac > $out/pmc_summary_prefix_H20_N65536_gradient_only.txt
echo "pmc (gradient only) done"
bash tools/pmc.sh $out/pmc_nz prof.py 20 65536 0 nz > $out/pmc_summary_prefix_H20_N65536_structured.txt
echo "pmc (structured dc/dx) done"
bash tools/pmc.sh $out/pmc_h100 prof.py 100 16384 0 jac > $out/pmc_summary_H100_N16384.txt
echo "pmc (H=100) done"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_qp -- python3 $R/tools/qp_time.py 4096 > $out/qp_solve_4096.txt 2> $out/kt_qp.log
echo "kernel trace QP done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_solve -- python3 $R/tools/resident_time.py 65536 1 6 --no-devqp > $out/solve_cfg3_65536_resident.txt 2> $out/kt_solve.log
echo "kernel trace full-size solve done"
cd $R
bash tools/pmc_qp.sh $out/pmc_qp 4096 > $out/pmc_summary_qp_kernel_4096.txt
echo "pmc (QP kernel) done"
