#!/bin/bash
# VERDICT r5 item 1, second pass: every op of the launch program issued 4 times in a row (CDC_DEV_REPEAT), kernel trace,
# first launch (cold) against launches 2..4 (hot) per kernel (tools/trace_cold_hot.py).  Output: gpurun_out/launch_floor/.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/launch_floor; mkdir -p $OUT
cd $R
timeout 300 tools/ubench/launch_floor > $OUT/ubench.txt 2>&1
grep -A8 "^## (h)" $OUT/ubench.txt
cd /tmp; export TMPDIR=/tmp
for B in 1 32; do
  S=24; [ $B = 32 ] && S=4
  rm -rf $OUT/rp
  CDC_DEV=1 CDC_DEV_REPEAT=4 CDC_NO_RANGE_GUARD=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/rp -o t -- python $R/bench.py --batch $B --sample-steps $S --prof-every 100000 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/rep_b${B}_stdout.txt 2>&1
  f=$(find $OUT/rp -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/trace_cold_hot.py $f 4 > $OUT/cold_hot_batch$B.txt 2>&1
  rm -rf $OUT/rp
  head -12 $OUT/cold_hot_batch$B.txt
done
