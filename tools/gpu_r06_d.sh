#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06d; mkdir -p $OUT
cd $R
export CDC_DEV=1
{
CDC_PW_DBG=1024 timeout 300 python tools/op_stress.py 32 192 64 64 384 1 1 0 150000
CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_pwprobe.so CDC_PW_DBG=9216 timeout 600 python tools/op_stress.py 32 192 64 64 384 1 1 0 400000
} 2>&1 | grep -v amdgpu.ids | sed "s#$R/##" | tee $OUT/pw_late_probe.txt
{
for rep in 1 2; do
python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "default (image-major narrow LayerNorm grids)"
CDC_LN_NO_IMG_MAJOR=1 python tools/gpu_b1_ab.py --batch 32 --sample-steps 40 --reps 2 --label "round-5 grid order"
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ln_grid_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "layernorm or weight_stationary or unet_forward_matches_reference_golden" 2>&1 | tail -4 | tee $OUT/pytest_ln.txt
