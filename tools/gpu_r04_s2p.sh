#!/bin/bash
# Development aid (round 4, second session): ddim_rows4_kernel -- bit identity against ddim_kernel, whole-model A/B, sampler tests.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2p; mkdir -p $O
timeout 600 python tools/gpu_ddim_rows4_check.py 2>&1 | grep -v amdgpu.ids | tee $O/check.txt
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
run() { echo -n "batch $2 $1: "; env CDC_DEV=1 $1 timeout 300 python bench.py --batch $2 $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']/100,4), 'ms/iter')"; }
{ for r in 1 2; do run "X=0" 32; run "CDC_NO_DDIM_ROWS4=1" 32; done; run "X=0" 1; run "CDC_NO_DDIM_ROWS4=1" 1; } 2>&1 | tee $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "decode or sampler or ddim or eps_param or full_resolution" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
