#!/bin/bash
# Development aid (round 4): ln_kernel_vec -- parity tests, then whole-model A/B of the 16-byte LayerNorm pass.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "layernorm or unet_forward or conv2d_matches or block_conv" > gpurun_out/ln_tests.log 2>&1
tail -4 gpurun_out/ln_tests.log
bash tools/gpu_toggles.sh "CDC_NO_LN_VEC=1" "CDC_LN_VEC_COLS=4" "CDC_LN_VEC_COLS=8" "CDC_LN_VEC_COLS=2" "CDC_NO_LN_VEC=1" 2>&1 | tee gpurun_out/ln_toggles.txt
CDC_BENCH_OPS=100 python bench.py --steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs 2>&1 >/dev/null | grep "^\[op\].* ln " > gpurun_out/ln_ops.txt
CDC_DEV=1 CDC_NO_LN_VEC=1 CDC_BENCH_OPS=100 python bench.py --steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs 2>&1 >/dev/null | grep "^\[op\].* ln " > gpurun_out/ln_ops_novec.txt
awk '{s+=$2} END{print "vec ln sum", s}' gpurun_out/ln_ops.txt; awk '{s+=$2} END{print "novec ln sum", s}' gpurun_out/ln_ops_novec.txt
