#!/bin/bash
# round 6: image-major grid of the one-launch context kernel (ctx_partial_kernel<., ONE>) against the round-5 order, ms per DDIM iteration + parity
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06f; mkdir -p $OUT
cd $R
export CDC_DEV=1
{
for rep in 1 2 3; do
python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "default (image-major context grids)"
CDC_CTX_NO_IMG_MAJOR=1 python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "round-5 grid order"
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ctx_grid_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "attention or unet_forward_matches_reference_golden or batch32_launch_plans" 2>&1 | tail -3 | tee -a $OUT/ctx_grid_ab.txt
