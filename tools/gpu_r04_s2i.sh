#!/bin/bash
# Development aid (round 4, second session): full GPU suite on the build with the column-major PF3 walk, the LDS-DMA fold kernels and the
# batched row-maxima loads; per-op ctx1 / ctxf times; whole-model ms per iteration at batch 32 / 1.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2i; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
for B in 32 1; do
  CDC_BENCH_OPS=400 timeout 300 python bench.py --batch $B $F 2> $O/err_b$B.txt | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('batch $B', round(j['ms_per_step']/100,4), 'ms/iter (per-op events on)')"
  grep "^\[op\]" $O/err_b$B.txt > $O/per_op_b$B.txt
  grep "ctx1\|ctxf" $O/per_op_b$B.txt | awk '{s+=$2; print} END{print "sum", s}'
  timeout 300 python bench.py --batch $B $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('batch $B', round(j['ms_per_step']/100,4), 'ms/iter')"
done
