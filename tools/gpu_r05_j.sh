#!/bin/bash
# Round 5: the launch list of one image per call (batch 1), per-op hipEvent times in program order
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_j; mkdir -p $OUT; cd $R
CDC_BENCH_OPS=400 CDC_BENCH_OPS_ORDER=1 timeout 300 python bench.py --batch 1 --sample-steps 100 --prof-every 2 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/bench_b1.json 2> $OUT/err_b1.txt
grep "^\[op\]" $OUT/err_b1.txt > $OUT/per_op_b1.txt
wc -l $OUT/per_op_b1.txt
python -c "
import json
d=json.loads(open('$OUT/bench_b1.json').read().strip().splitlines()[-1]); print('batch 1 ms/iter', d['roofline']['ms_per_ddim_iter'])"
