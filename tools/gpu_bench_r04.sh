#!/bin/bash
# The default bench line with the per-op table (part of tools/gpu_profiles_r04.sh, for a re-run of that part alone)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles_r04; mkdir -p $OUT; cd $R
CDC_BENCH_OPS=400 timeout 1200 python bench.py > $OUT/bench_r04.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op_r04.txt
tail -1 $OUT/bench_r04.json | cut -c1-300
