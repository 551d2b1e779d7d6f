#!/bin/bash
# Development aid (round 4, second session): forced launch plans (CDC_PLAN=MB,NPW,0) on the <= 16^2 Block shapes at batch 32 and batch 1
# (convolution + LayerNorm pass, tools/gpu_conv_tune.py).
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2d; mkdir -p $O
for B in 32 1; do
for shp in "384 8 8 384" "320 16 16 320" "256 16 16 256" "768 8 8 320" "640 16 16 256"; do
  set -- $shp
  timeout 600 python tools/gpu_conv_tune.py $B $1 $2 $3 $4 3 1 1 auto 1,1,0 2,1,0 1,2,0 2,2,0 3,1,0 4,1,0 3,2,0 4,2,0 5,1,0 6,1,0 2>&1 | grep -v "^$"
done
done | tee $O/trunk_plans.txt
