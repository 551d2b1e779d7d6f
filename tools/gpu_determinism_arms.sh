#!/bin/bash
# Round 6: arms of the localisation of conv_pw_kernel's counted-wait event on ONE box (the rate differs between boxes).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/determinism; mkdir -p $OUT
cd $R
N=${1:-400000}
{
CDC_PW_DBG=1024 timeout 900 python tools/op_stress.py 32 192 64 64 384 1 1 0 $N
CDC_PW_DBG=3072 timeout 900 python tools/op_stress.py 32 192 64 64 384 1 1 0 $N
CDC_PW_DBG=5120 timeout 900 python tools/op_stress.py 32 192 64 64 384 1 1 0 $N
CDC_PW_DBG=1024 CDC_NO_PW_X16=1 timeout 900 python tools/op_stress.py 32 192 64 64 384 1 1 0 $N
CDC_PW_DBG=1024 timeout 900 python tools/op_stress.py 32 192 64 64 384 1 1 0 $N
} 2>&1 | grep -v amdgpu.ids | tee $OUT/pw_arms.txt
timeout 600 tools/ubench/dma_order 100000 16 scatter 2>&1 | tee $OUT/dma_order_scatter.txt
