#!/bin/bash
# Development aid (round 4, second session): conv_pf3_kernel tile walk order (CDC_PF3_XMAJOR=1: row-major) -- whole-model A/B, per-op tables,
# parity of the batch-32 paths under the switch.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2g; mkdir -p $O
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
run() { echo -n "$1: "; env CDC_DEV=1 $1 timeout 300 python bench.py $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(j['ms_per_step']/100,4), 'ms/iter', 'pf3', round(r['families']['conv_pf3_kernel']['ms_per_iteration'],3), 'dominant', round(r['avg_launch_ms'],4))"; }
{ for r in 1 2 3; do run "X=0"; run "CDC_PF3_XMAJOR=1"; done; } 2>&1 | tee $O/ab.txt
CDC_DEV=1 CDC_PF3_XMAJOR=1 CDC_BENCH_OPS=400 timeout 300 python bench.py $F 2>&1 >/dev/null | grep "^\[op\].*PF3" > $O/per_op_ymajor.txt
CDC_BENCH_OPS=400 timeout 300 python bench.py $F 2>&1 >/dev/null | grep "^\[op\].*PF3" > $O/per_op_xmajor.txt
paste <(awk '{print $2}' $O/per_op_xmajor.txt) <(awk '{print $2}' $O/per_op_ymajor.txt) <(cut -c30- $O/per_op_xmajor.txt) | head -20
CDC_DEV=1 CDC_PF3_XMAJOR=1 timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "configs1 or full_resolution or batch32 or eps_param_256 or x_param_512 or unet_forward_matches" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
