#!/bin/bash
# Round-4: row-layout epilogue accesses of conv_pf_kernel -- tests, then the per-op bench with them (default) and without (CDC_PF_DBG=512)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04s; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "pre_split or stride2 or fused_phases or stage_taps or alternate_kernel_modes or pointwise or persistent or context_decoder or compressor_forward" > $OUT/pytest_new.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_new.log
run_bench() {
    tag=$1; shift
    env "$@" CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    grep "^\[op\]" $OUT/bench_$tag.err > $OUT/per_op_$tag.txt
    python3 -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), 'verify', d.get('verify',{}).get('max_rel_err_vs_batch1_decode'), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()}, {k:round(v['ms_per_iteration'],3) for k,v in d['roofline']['families'].items()})"
}
run_bench rows CDC_X=0
run_bench lanes CDC_DEV=1 CDC_PF_DBG=512
run_bench rows2 CDC_X=0
