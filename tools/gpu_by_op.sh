#!/bin/bash
# per-op / per-level time of one DDIM iteration without hipEvent pairs in the stream (tools/trace_by_op.py), batch 32 and batch 1
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/by_op; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for B in 32 1; do
  S=8; [ $B = 1 ] && S=40
  rm -rf $OUT/rp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/rp -o t -- python $R/bench.py --batch $B --sample-steps $S --prof-every 100000 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras --dump-ops $OUT/ops_b$B.txt > $OUT/stdout_b$B.txt 2>&1
  f=$(find $OUT/rp -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/trace_by_op.py $f $OUT/ops_b$B.txt --batch $B > $OUT/by_op_batch$B.txt 2>&1
  rm -rf $OUT/rp
  head -6 $OUT/by_op_batch$B.txt
done
