#!/bin/bash
# Development aid (round 4, second session): six K slices at batch 1 (ln_kernel_vec<2, 5|6>): full GPU suite, batch-1 / batch-32 ms per
# iteration, batch-1 per-class sums.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2m; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
for cfg in "X=0" "CDC_KMAX=4" "X=0" "CDC_KMAX=4"; do
  echo -n "batch 1 $cfg: "; env CDC_DEV=1 $cfg timeout 300 python bench.py --batch 1 $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']/100,4), 'ms/iter')"
done 2>&1 | tee $O/ab_b1.txt
CDC_BENCH_OPS=400 timeout 300 python bench.py --batch 1 $F 2>&1 >/dev/null | grep "^\[op\]" > $O/per_op_b1.txt
grep "ln C=384\|384->384  out   8x8" $O/per_op_b1.txt | head -6
echo -n "batch 2: "; timeout 300 python bench.py --batch 2 $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']/100,4), 'ms/iter')"
