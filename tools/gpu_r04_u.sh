#!/bin/bash
# Round-4: one-launch attention context at the few-pixel levels -- attention tests, U-Net tests, per-op bench with / without
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04u; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "attention or unet_forward or taps or batch32 or digest" > $OUT/pytest_new.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_new.log
run_bench() {
    tag=$1; shift
    env "$@" CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    grep "^\[op\]" $OUT/bench_$tag.err > $OUT/per_op_$tag.txt
    python3 -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), 'verify', d.get('verify',{}).get('max_rel_err_vs_batch1_decode'), 'batch1', d.get('batch1',{}).get('ms_per_ddim_iter'), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()})"
}
run_bench new CDC_X=0
run_bench old CDC_DEV=1 CDC_NO_CTX_ONE=1
grep -E "ctx1|ctxp|ctxr|kstats" $OUT/per_op_new.txt | head -12
