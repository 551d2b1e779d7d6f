#!/bin/bash
# Round 6: long stress of the plane-operand kernels' counted waits at the layer shapes of BASELINE configs[1] (batch 32), and the
# dma_order ubench with LDS read pressure.  Output: gpurun_out/determinism/
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/determinism; mkdir -p $OUT
cd $R
N=${1:-200000}
export CDC_PF=1 CDC_PF_MAXPIX=0 CDC_WS_MIN_WGS=1000000000
{
timeout 900 python tools/op_stress.py 32 64 256 256 64 3 1 1 $((N/2))
timeout 900 python tools/op_stress.py 32 128 128 128 128 3 1 1 $((N/2))
timeout 900 python tools/op_stress.py 32 192 64 64 192 3 1 1 $N
timeout 900 python tools/op_stress.py 32 256 32 32 256 3 1 1 $N
timeout 900 python tools/op_stress.py 32 64 256 256 64 3 2 1 $N
timeout 900 python tools/op_stress.py 32 128 128 128 128 3 2 1 $N
timeout 900 python tools/op_stress.py 32 384 64 64 128 1 1 0 $N
} 2>&1 | grep -v amdgpu.ids | tee $OUT/pf_long.txt
timeout 900 tools/ubench/dma_order 100000 16 scatter 2>&1 | tee $OUT/dma_order_scatter.txt
