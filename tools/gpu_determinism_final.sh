#!/bin/bash
# Round 6, final build: box indicator (conv_pw_kernel's counted wait), then the whole 500-step batch-32 decode again and again, bits compared
# (tools/determinism_stress.py), then every DDIM step repeated from the reference trajectory (tools/determinism_stress_steps.py).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/determinism; mkdir -p $OUT
cd $R
{
CDC_DEV=1 CDC_PW_DBG=1024 timeout 300 python tools/op_stress.py 32 192 64 64 384 1 1 0 150000
timeout 1200 python tools/determinism_stress.py 100 40 2>&1 | tail -4
CDC_DEV=1 timeout 1200 python tools/determinism_stress_steps.py 40 2>&1 | grep -v "^taps" | cut -c1-300 | tail -4
} 2>&1 | grep -v amdgpu.ids | tee $OUT/final_build.txt
