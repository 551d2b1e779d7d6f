#!/bin/bash
# Round 6, third pass: late-piece probe of conv_pw_kernel (lab build), per-op times again (clean iterations only)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06c; mkdir -p $OUT
cd $R
{
CDC_DEV=1 CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_pwprobe.so CDC_PW_DBG=9216 timeout 600 python tools/op_stress.py 32 192 64 64 384 1 1 0 300000
} 2>&1 | grep -v amdgpu.ids | sed "s#$R/##" | tee $OUT/pw_late_probe.txt
bash tools/gpu_by_op.sh > $OUT/by_op_log.txt 2>&1
cp gpurun_out/by_op/by_op_batch*.txt $OUT/
