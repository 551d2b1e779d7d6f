#!/bin/bash
# Development aid (round 4, second session): splits of the fused attention front half at C = 128 (CDC_KV128_WGS) and C = 64 (CDC_KV64_WGS)
# with the faster fold: whole-model ms per iteration + attention class ms, batch 32.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2q; mkdir -p $O
F="--steps 1 --warmup 1 --sample-steps 60 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs --prof-every 10"
run() { echo -n "$1: "; env CDC_DEV=1 $1 timeout 300 python bench.py $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; c=r['class_ms_per_ddim_iter']
print(round(r['ms_per_ddim_iter'],3), 'ms/iter', 'attn', round(c['attn_ctx'],3))"; }
{ run "X=0"; for t in "$@"; do run "$t"; done; run "X=0"; } 2>&1 | tee $O/kv_splits.txt
