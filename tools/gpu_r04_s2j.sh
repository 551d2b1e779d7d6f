#!/bin/bash
# Development aid (round 4, second session): register-block shapes (MB, NPW) of the split-K convolutions at the 8x8 / 16x16 levels, forced
# in the whole model (CDC_PLAN8 / CDC_PLAN16 = "MB,NPW,0"); ms per DDIM iteration at batch 32 (60-iteration decodes) + conv3x3 / conv1x1 class ms.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2j; mkdir -p $O
F="--steps 1 --warmup 1 --sample-steps 60 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs --prof-every 10"
run() { echo -n "$1: "; env CDC_DEV=1 $1 timeout 300 python bench.py $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; c=r['class_ms_per_ddim_iter']
print(round(r['ms_per_ddim_iter'],3), 'ms/iter', 'conv3x3', round(c['conv3x3'],3), 'conv1x1', round(c['conv1x1'],3), 'ln', round(c['layernorm'],3), 'split2', round(r['families']['conv_split2_kernel']['ms_per_iteration'],3))"; }
{ run "X=0"; for t in "$@"; do run "$t"; done; run "X=0"; } 2>&1 | tee $O/plans.txt
