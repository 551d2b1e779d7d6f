#!/bin/bash
# what a dependent launch boundary costs.  (1) the ubench sweep, (2) the same sweep's chain under
# rocprofv3 --kernel-trace (profiler timestamps next to the device stamps), (3) the gaps between the LIBRARY's own launches
# at one image per call and at batch 32 (tools/trace_gaps.py).  Output: gpurun_out/launch_floor/.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/launch_floor; mkdir -p $OUT
cd $R
timeout 300 tools/ubench/launch_floor > $OUT/ubench.txt 2>&1
tail -5 $OUT/ubench.txt
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/rp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/rp -o u -- $R/tools/ubench/launch_floor quick > $OUT/ubench_under_rocprof.txt 2>&1
f=$(find $OUT/rp -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/trace_gaps.py $f --break-us 30 > $OUT/ubench_trace_gaps.txt 2>&1
rm -rf $OUT/rp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/rp -o b1 -- python $R/bench.py --batch 1 --sample-steps 60 --prof-every 100000 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/b1_stdout.txt 2>&1
f=$(find $OUT/rp -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/trace_gaps.py $f --break-us 30 > $OUT/batch1_trace_gaps.txt 2>&1
rm -rf $OUT/rp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/rp -o b32 -- python $R/bench.py --batch 32 --sample-steps 6 --prof-every 100000 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/b32_stdout.txt 2>&1
f=$(find $OUT/rp -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $R/tools/trace_gaps.py $f --break-us 30 > $OUT/batch32_trace_gaps.txt 2>&1
rm -rf $OUT/rp
head -8 $OUT/batch1_trace_gaps.txt; head -8 $OUT/batch32_trace_gaps.txt
