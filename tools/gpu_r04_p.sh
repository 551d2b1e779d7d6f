#!/bin/bash
# Round-4: fused-phase transposed kernel -- ablation (variant pfabl) and the one-workgroup-per-CU build (variant tz1wg)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04p; mkdir -p $OUT; cd $R
export CDC_DEV=1
for dbg in 0 16 256 271 367; do
    CDC_NO_RANGE_GUARD=1 CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_pfabl.so CDC_PF_DBG=$dbg CDC_BENCH_OPS=400 python bench.py --sample-steps 30 --prof-every 5 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs --no-verify > $OUT/bench_$dbg.json 2> $OUT/bench_$dbg.err
    echo "dbg=$dbg $(grep -E '^\[op\].* TZ4' $OUT/bench_$dbg.err | awk '{printf "%s(%s) ", $2, $8}')"
done
for v in "" tz1wg; do
    lib=$R/cdc_compression_amd/libcdc_hip${v:+_$v}.so
    CDC_HIP_LIB=$lib CDC_BENCH_OPS=400 python bench.py --sample-steps 60 --prof-every 5 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs --no-verify > $OUT/bench_v$v.json 2> $OUT/bench_v$v.err
    echo "variant=${v:-default} $(grep -E '^\[op\].* TZ4' $OUT/bench_v$v.err | awk '{printf "%s(%s) ", $2, $8}')"
done
