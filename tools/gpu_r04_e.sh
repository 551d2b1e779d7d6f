#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04e; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
bash tools/gpu_toggles.sh "CDC_PF_JOIN_MAXPIX=4096" "CDC_NO_PF_ONLY_JOIN=1" 2>&1 | tee $OUT/toggles.txt
CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > $OUT/bench.json 2> $OUT/bench.err
grep "^\[op\]" $OUT/bench.err > $OUT/per_op.txt; head -30 $OUT/per_op.txt
