#!/bin/bash
# Round 5: which conv_ws1_kernel launches make the batch-32 decode differ between runs?  (launches below the threshold stay on the round-4 kernels)
set -u
R=$GRAFT_REPO_ROOT; cd $R
for v in 200 350 500 1100; do
echo "== CDC_WS1_MIN_WGS=$v"
env CDC_DEV=1 CDC_WS1_MIN_WGS=$v timeout 600 python tools/determinism_stress.py 24 0 2>&1 | grep -v amdgpu.ids | tail -6
done
