#!/bin/bash
# Round-4 experiment: variant libraries against the default, per-op lines matching a pattern.  usage: gpu_r04_y.sh PATTERN variant...
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04y; mkdir -p $OUT; cd $R
PAT="$1"; shift
for v in "" "$@"; do
    lib=$R/cdc_compression_amd/libcdc_hip${v:+_$v}.so
    CDC_HIP_LIB=$lib CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_v$v.json 2> $OUT/bench_v$v.err
    echo "variant=${v:-default} $(python3 -c "
import json; d=json.loads(open('$OUT/bench_v$v.json').read().strip().splitlines()[-1]); print(round(d['roofline']['ms_per_ddim_iter'],3), d['verify']['max_rel_err_vs_batch1_decode'])")"
    grep -E "$PAT" $OUT/bench_v$v.err | cut -c1-140 | head -6
done
