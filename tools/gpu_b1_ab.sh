#!/bin/bash
# unprofiled A/B at one image per call and at batch 32: repeat slope (is the profiler's per-kernel duration the real cost?),
# null stream vs a non-blocking side stream, eager vs hipGraph.  Output: gpurun_out/launch_floor/b1_ab.txt
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/launch_floor; mkdir -p $OUT
cd $R
{
for rep in 1 2 3; do CDC_DEV=1 CDC_DEV_REPEAT=$rep CDC_NO_RANGE_GUARD=1 python tools/gpu_b1_ab.py --batch 1 --sample-steps 200; done
python tools/gpu_b1_ab.py --batch 1 --sample-steps 200 --side-stream
CDC_GRAPH=1 python tools/gpu_b1_ab.py --batch 1 --sample-steps 200
CDC_GRAPH=1 python tools/gpu_b1_ab.py --batch 1 --sample-steps 200 --side-stream
for rep in 1 2; do CDC_DEV=1 CDC_DEV_REPEAT=$rep CDC_NO_RANGE_GUARD=1 python tools/gpu_b1_ab.py --batch 32 --sample-steps 20 --reps 2; done
python tools/gpu_b1_ab.py --batch 32 --sample-steps 20 --reps 2 --side-stream
} 2>&1 | grep -v "^$" | tee $OUT/b1_ab.txt
