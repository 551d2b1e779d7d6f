#!/bin/bash
# Development aid (round 4, second session): 192-channel 3x3 layer at 64x64 / batch 32 on conv_pf_kernel: 8-row tiles (512 workgroups, the
# planner's choice) against 4-row tiles (1024 workgroups), with and without a residual.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2k; mkdir -p $O
export CDC_DEV=1 CDC_PF=1 CDC_PF_MAXPIX=0
{
for res in "" 1; do
  echo "--- residual: ${res:-no}"
  TUNE_RESID=$res timeout 300 python tools/gpu_conv_tune.py 32 192 64 64 192 3 1 1 auto | grep -v "^\[plan\]" | tail -2
  CDC_PF_PLAN=3,1,2,4 TUNE_RESID=$res timeout 300 python tools/gpu_conv_tune.py 32 192 64 64 192 3 1 1 auto | grep -v "^\[plan\]" | tail -2
  CDC_PF_PLAN=3,2,2,4 TUNE_RESID=$res timeout 300 python tools/gpu_conv_tune.py 32 192 64 64 192 3 1 1 auto | grep -v "^\[plan\]" | tail -2
done
} 2>&1 | tee $O/pf192.txt
