#!/bin/bash
# Development aid (round 4, second session): six against four K slices at small batches (ms per DDIM iteration, 100-iteration decodes).
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2n; mkdir -p $O
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
for B in ${BATCHES:-2 3 4 8}; do for K in 4 6 4 6; do
  echo -n "batch $B CDC_KMAX=$K: "; env CDC_DEV=1 CDC_KMAX=$K timeout 300 python bench.py --batch $B $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']/100,4), 'ms/iter')"
done; done 2>&1 | tee $O/kmax_small_batches.txt
