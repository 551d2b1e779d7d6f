#!/bin/bash
# Round 6, VERDICT r5 item 2.  Output: gpurun_out/determinism/
#  1. tools/ubench/dma_order: is a counted s_waitcnt vmcnt exact for LDS-DMA pieces (issue-order retirement)?
#  2. the layer of round 5's event (1x1 192 -> 384 @64x64, batch 32, three channel groups) on conv_pw_kernel: counted wait vs vmcnt(0),
#     200 000 executions each, bitwise against the first (cdc_op_stress)
#  3. cost of -DCDC_DMA_WAIT_ALL (every counted wait of conv_pf_kernel / conv_pf3_kernel -> vmcnt(0)) at batch 32 and batch 1
#  4. the long-form pytest guard
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/determinism; mkdir -p $OUT
cd $R
timeout 300 tools/ubench/dma_order 20000 8 > $OUT/dma_order.txt 2>&1
{
CDC_PW_COUNTED_WAIT=1 timeout 600 python tools/op_stress.py 32 192 64 64 384 1 1 0 200000
timeout 600 python tools/op_stress.py 32 192 64 64 384 1 1 0 200000
} 2>&1 | grep -v amdgpu.ids | tee $OUT/pw_event_layer.txt
{
for rep in 1 2; do
python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "default build (counted waits)"
CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_waitall.so python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "-DCDC_DMA_WAIT_ALL build"
done
python tools/gpu_b1_ab.py --batch 1 --sample-steps 200 --label "default build (counted waits)"
CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_waitall.so python tools/gpu_b1_ab.py --batch 1 --sample-steps 200 --label "-DCDC_DMA_WAIT_ALL build"
} 2>&1 | grep -v amdgpu.ids | tee $OUT/wait_all_ab.txt
timeout 900 python -m pytest tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_determinism.txt
