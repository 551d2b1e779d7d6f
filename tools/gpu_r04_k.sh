#!/bin/bash
# Round-4 check of a conv_split2_kernel change: the whole GPU suite, the long determinism screen, a short bench with the per-op table.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04k; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python tools/gpu_determinism_long.py > $OUT/determinism.log 2>&1; echo "determinism rc=$?"; tail -3 $OUT/determinism.log
CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > $OUT/bench.json 2> $OUT/bench.err
grep "^\[op\]" $OUT/bench.err > $OUT/per_op.txt
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()}, {k:round(v['ms_per_iteration'],3) for k,v in d['roofline']['families'].items()})"
grep -E "s2 |2x2 s1|1x7|7x1" $OUT/per_op.txt | head -12
