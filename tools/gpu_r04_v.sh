#!/bin/bash
# Round-4: conv_pf3_kernel with the non-synchronised epilogue schedule for the `nof32 +pf` variant (variant nosync4) against the default
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04v; mkdir -p $OUT; cd $R
run_bench() {
    tag=$1; shift
    env "$@" CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    grep "^\[op\]" $OUT/bench_$tag.err > $OUT/per_op_$tag.txt
    python3 -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), 'verify', d.get('verify',{}).get('max_rel_err_vs_batch1_decode'), {k:round(v['ms_per_iteration'],3) for k,v in d['roofline']['families'].items()})"
    grep -E "PF3 LN nof32 \+pf$" $OUT/per_op_$tag.txt | head -8
}
run_bench sync CDC_X=0
run_bench nosync CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_nosync4.so
