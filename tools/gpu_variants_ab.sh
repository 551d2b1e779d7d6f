#!/bin/bash
# Round 6, second measurement pass on ONE box: event-rate indicator of the box (conv_pw_kernel's counted wait), per-op / per-level times
# without event pairs, A/B of the build variants (write-through stores of conv_pf3_kernel, wait margins), the long-form determinism test,
# the long stress of the plane-operand kernels.
set -u
# (build the variants first: tools/build_variant.sh pf3wt -DCDC_PF3_ST_WT; margin -DCDC_DMA_WAIT_MARGIN=1 [=0 is the round-5 form now]; margin6 "... -DCDC_PF3_D=6")
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06b; mkdir -p $OUT
cd $R
CDC_DEV=1 CDC_PW_DBG=1024 timeout 300 python tools/op_stress.py 32 192 64 64 384 1 1 0 150000 2>&1 | grep -v amdgpu.ids | tee $OUT/box_indicator.txt
bash tools/gpu_by_op.sh > $OUT/by_op_log.txt 2>&1
cp gpurun_out/by_op/by_op_batch*.txt gpurun_out/by_op/ops_b*.txt $OUT/ 2>/dev/null
{
for rep in 1 2; do
python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "default build"
for v in pf3wt margin margin6; do
CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_$v.so python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "variant $v"
done
done
} 2>&1 | grep -v amdgpu.ids | sed "s#$R/##" | tee $OUT/variants_ab.txt
timeout 900 python -m pytest tests/test_gpu_determinism.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_determinism.txt
bash tools/gpu_determinism_long.sh 100000 > $OUT/long_log.txt 2>&1
cp gpurun_out/determinism/pf_long.txt $OUT/pf_long.txt
