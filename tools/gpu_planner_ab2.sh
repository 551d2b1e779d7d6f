#!/bin/bash
# round 6: 1x1 layers of the 16 x 16 level at batch 32 on conv_ws1_kernel where they have few channel groups (A/B), ms per DDIM iteration
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06e; mkdir -p $OUT
cd $R
export CDC_DEV=1
{
for rep in 1 2; do
python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "default (conv_ws1_kernel up to 128 pixel blocks)"
CDC_WS1_MAX_GROUPS=8 python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "up to 256 blocks where <= 8 channel groups"
CDC_WS1_MAX_GROUPS=10 python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "up to 256 blocks where <= 10 channel groups"
CDC_WS1_MAX_BLOCKS=256 python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "up to 256 blocks, any width"
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ws1_ab.txt
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4 | tee $OUT/pytest.txt
