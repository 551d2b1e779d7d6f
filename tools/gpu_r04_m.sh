#!/bin/bash
# Round-4: stride-2 plane-operand kernel, the default build against the variant libraries named on the command line; stride-2 op tests first.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04m; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "stride2 or planes_only or stage_taps or fused_phases or alternate_kernel_modes or conv_transpose" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -3 $OUT/pytest_new.log
run_bench() {   # tag, env...
    tag=$1; shift
    env "$@" CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    grep "^\[op\]" $OUT/bench_$tag.err > $OUT/per_op_$tag.txt
    python3 -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), 'verify', d.get('verify',{}).get('max_rel_err_vs_batch1_decode'), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()})"
    grep -E " s2 | 2x2 " $OUT/per_op_$tag.txt | head -8
}
run_bench default CDC_X=0
for extra in "$@"; do run_bench "$extra" CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_$extra.so; done
