#!/bin/bash
# Round-4: stride-2 convolutions on the plane-operand kernel -- new tests first, then the whole suite, then the per-op bench with
# and without the new path.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04l; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "stride2 or planes_only or alternate_kernel_modes or stage_taps" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -6 $OUT/pytest_new.log
run_bench() {   # tag, env...
    tag=$1; shift
    env "$@" CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    grep "^\[op\]" $OUT/bench_$tag.err > $OUT/per_op_$tag.txt
    python3 -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), 'verify', d.get('verify'), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()})"
    grep -E " s2 |1x1 s1   64->64   out 256" $OUT/per_op_$tag.txt | head -8
}
run_bench new CDC_X=0
run_bench old CDC_DEV=1 CDC_NO_PF_S2=1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
