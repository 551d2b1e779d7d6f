#!/bin/bash
# Round-4: environment-switch A/B of the default build.  usage: gpu_r04_z.sh PATTERN "ENV=1 ENV2=2" ...
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04z; mkdir -p $OUT; cd $R
PAT="$1"; shift
i=0
for e in "CDC_X=0" "$@"; do
    env CDC_DEV=1 $e CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$i.json 2> $OUT/bench_$i.err
    echo "[$e] $(python3 -c "
import json; d=json.loads(open('$OUT/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['roofline']['ms_per_ddim_iter'],3), d['verify']['max_rel_err_vs_batch1_decode'], {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()})")"
    grep -E "$PAT" $OUT/bench_$i.err | cut -c1-120 | head -6
    i=$((i+1))
done
