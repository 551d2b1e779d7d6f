#!/bin/bash
# Development aid (round 4, second session): batch-1 sweep of the planner thresholds (ms per DDIM iteration, 100-iteration decodes), full
# per-op table at batch 1.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2c; mkdir -p $O
F="--batch ${BATCH:-1} --steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
run() { echo -n "$1: "; env CDC_DEV=1 $1 timeout 200 python bench.py $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']/100,4), 'ms/iter')"; }
[ -n "$SKIP_OPS" ] || CDC_BENCH_OPS=400 timeout 300 python bench.py $F 2>&1 >/dev/null | grep "^\[op\]" > $O/per_op_b1_full.txt
wc -l $O/per_op_b1_full.txt
{
run "X=0"
for t in "$@"; do run "$t"; done
run "X=0"
} 2>&1 | tee $O/sweep_b1.txt
