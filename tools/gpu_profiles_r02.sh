#!/bin/bash
# Collects the round-2 evidence files (copy gpurun_out/profiles_r02/* to profiles/):
#   bench_r02.json               the default bench line
#   per_op_r02.txt               per-op hipEvent averages of the same run (CDC_BENCH_OPS)
#   rocprof_r02_kernel_stats.csv rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline`
#   pmc_r02_*                    PMC passes for the dominant launch shape (tools/gpu_pmc_bench_traffic.sh)
set -u
export CDC_DEV=1      # the CDC_* planner switches below are development switches (cdc_internal.h: dev_env)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles_r02; mkdir -p $OUT
cd $R
CDC_BENCH_OPS=400 python bench.py > $OUT/bench_r02.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op_r02.txt
tail -1 $OUT/bench_r02.json | cut -c1-600
( cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/rp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp -o k -- python $R/bench.py --no-cpu-baseline --no-verify > $OUT/rocprof_bench_stdout.txt 2>&1 )
f=$(find $OUT/rp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_r02_kernel_stats.csv
rm -rf $OUT/rp
head -8 $OUT/rocprof_r02_kernel_stats.csv | cut -c1-200
CDC_PF=1 TUNE_RESID=1 bash tools/gpu_pmc_bench_traffic.sh "32 64 256 256 64 3 1 1" > $OUT/pmc_log.txt 2>&1
cp $R/gpurun_out/pmc_r02/pmc_r02_* $OUT/ 2>/dev/null
tail -5 $OUT/pmc_log.txt | cut -c1-400
