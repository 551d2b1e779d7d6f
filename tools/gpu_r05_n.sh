#!/bin/bash
# Round 5: launch list of one image per call (program order, per-op times)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_n; mkdir -p $OUT; cd $R
CDC_BENCH_OPS=400 CDC_BENCH_OPS_ORDER=1 timeout 300 python bench.py --batch 1 --sample-steps 100 --prof-every 2 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/b1.json 2> $OUT/err_b1.txt
grep "^\[op\]" $OUT/err_b1.txt > $OUT/per_op_b1.txt
wc -l $OUT/per_op_b1.txt
