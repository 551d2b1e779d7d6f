#!/bin/bash
# Round 5: small batches -- the plane-operand stride-2 / fused-phase kernels from fewer workgroups than one per CU
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_o; mkdir -p $OUT; cd $R
for v in "CDC_X_UNUSED=1" "CDC_PF_S2_MIN_WGS=32" "CDC_PF_TZ_MIN_WGS=32" "CDC_PF_S2_MIN_WGS=32 CDC_PF_TZ_MIN_WGS=32" "CDC_PF_S2_MIN_WGS=128 CDC_PF_TZ_MIN_WGS=128"; do
for b in 1 2 4; do
env CDC_DEV=1 $v timeout 300 python bench.py --batch $b --sample-steps 60 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] batch $b: ms/iter', d['roofline']['ms_per_ddim_iter'], d.get('batch1'))"
done; done
