#!/bin/bash
# Development aid (round 4, second session): batch-1 per-op table, hipGraph replay re-measured at batch 1 / 4 / 32, two half-batches on two streams.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2b; mkdir -p $O
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
CDC_BENCH_OPS=50 timeout 300 python bench.py --batch 1 $F > $O/b1_ops.json 2> $O/b1_ops_stderr.txt
grep "^\[op\]" $O/b1_ops_stderr.txt > $O/per_op_b1.txt
for B in 1 4 32; do
  for G in 0 1; do
    CDC_GRAPH=$G timeout 300 python bench.py --batch $B $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch $B graph $G ms/iter', d['roofline']['ms_per_ddim_iter'] if 'roofline' in d else d['ms_per_step']/100, 'value', d['value'])"
  done
done 2>&1 | tee $O/graph_ab.txt
timeout 300 python tools/gpu_two_streams.py 60 2 2>&1 | tail -1 | tee $O/two_streams.txt
timeout 300 python tools/gpu_two_streams.py 60 2 2>&1 | tail -1 | tee -a $O/two_streams.txt
