#!/bin/bash
# planner-threshold A/Bs of the few-pixel levels on one box (ms per DDIM iteration, unprofiled): 128-pixel tiles from 160 workgroups,
# stride-2 layers on the weight-stationary kernel at batch 32, the stride-1 upper bound at mid-size batches
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06b; mkdir -p $OUT
cd $R
export CDC_DEV=1
{
for rep in 1 2; do
python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "default"
CDC_WS_NPB=4 python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "128-pixel tiles everywhere"
CDC_WS_MIN_WGS=4 python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "stride-2 gate lifted"
done
for B in 4 8 16; do
python tools/gpu_b1_ab.py --batch $B --sample-steps 60 --reps 2 --label "no upper bound"
CDC_WS_MAX_WGS=1280 python tools/gpu_b1_ab.py --batch $B --sample-steps 60 --reps 2 --label "stride-1 launches <= 1280 workgroups"
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/planner_ab.txt
