#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04d; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "conv2d or unet_forward_matches_reference" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/gpu_toggles.sh "CDC_PF_JOIN_MAXPIX=16384" "CDC_PF_JOIN_MAXPIX=16384 CDC_PF_TRANSPOSED=1" "CDC_PF_TRANSPOSED=1" "CDC_SPLIT2_PIPE=0" 2>&1 | tee $OUT/toggles.txt
for cfg in base join16k_T; do
  env="CDC_DEV=1"; [ $cfg = join16k_T ] && env="CDC_DEV=1 CDC_PF_JOIN_MAXPIX=16384 CDC_PF_TRANSPOSED=1"
  env $env CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  grep "^\[op\]" $OUT/bench_$cfg.err > $OUT/per_op_$cfg.txt
done
