#!/bin/bash
# Round-4: where the time of the stride-2 plane-operand kernel goes -- ablation switches (CDC_PF_DBG bits, wrong results) on the
# -DCDC_PF_ABLATE=1 variant; prints the per-op time of the two big Downsample layers per setting.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04n; mkdir -p $OUT; cd $R
export CDC_DEV=1 CDC_NO_RANGE_GUARD=1 CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_pfabl.so
for dbg in 0 1 2 3 4 8 15 32 64 96 16 256 271 367; do
    CDC_PF_DBG=$dbg CDC_BENCH_OPS=400 python bench.py --sample-steps 30 --prof-every 5 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs --no-verify > $OUT/bench_$dbg.json 2> $OUT/bench_$dbg.err
    echo "dbg=$dbg $(grep -E '^\[op\].*( s2 +(64->64|128->128) | TZ4)' $OUT/bench_$dbg.err | awk '{printf "%s(%s %s) ", $2, $7, $8}')"
done
