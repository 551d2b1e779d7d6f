#!/bin/bash
# Round-4: ablation builds of conv_split2_kernel on the stride-2 shapes (what bounds them?), plus a check of the pre_init change.
set -u
R=$GRAFT_REPO_ROOT; [ -f $GRAFT_REPO_ROOT/cdc_compression_amd/libcdc_hip_timeline.so ] || bash $GRAFT_REPO_ROOT/tools/build_variant.sh timeline "-DCDC_TIMELINE" > /dev/null 2>&1; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "unet_forward or decode_matches_reference or stage_taps or context_decoder or pre_split" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in NOSTOREX NOLOADX NOW NOMFMA NOEPI; do [ -f $R/cdc_compression_amd/libcdc_hip_ab_$v.so ] || bash $R/tools/build_variant.sh ab_$v "-DCDC_AB_$v" > /dev/null 2>&1; done
export CDC_DEV=1 TUNE_CHILD=1
for SHAPE in "32 64 256 256 64 3 2 0" "32 128 128 128 128 3 2 0" "32 64 256 256 64 3 1 0"; do
  echo "== shape $SHAPE"
  for v in main ab_NOSTOREX ab_NOLOADX ab_NOW ab_NOMFMA ab_NOEPI; do
    lib=$R/cdc_compression_amd/libcdc_hip_$v.so; [ $v = main ] && lib=$R/cdc_compression_amd/libcdc_hip.so
    echo -n "$v pipe0: "; CDC_HIP_LIB=$lib CDC_SPLIT2_PIPE=0 timeout 120 python tools/gpu_conv_tune.py $SHAPE 2>&1 | tail -1
  done
  echo -n "main pipe1: "; CDC_SPLIT2_PIPE=1 timeout 120 python tools/gpu_conv_tune.py $SHAPE 2>&1 | tail -1
done
CDC_BENCH_OPS=400 timeout 600 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > $OUT/bench.json 2> $OUT/bench.err
grep "^\[op\]" $OUT/bench.err > $OUT/per_op.txt; grep "HOIST\|7x1\|PF LN nof32 +pf" $OUT/per_op.txt | head -12
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value %.3f  ms/iter %.3f' % (d['value'], d['roofline']['ms_per_ddim_iter']), {k:round(v['ms_per_iteration'],3) for k,v in d['roofline']['families'].items()})"
for mode in 0 1; do
CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_timeline.so CDC_SPLIT2_PIPE=$mode timeout 300 python bench.py --sample-steps 2 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > /dev/null 2> $OUT/timeline_pipe$mode.err
grep "^\[timeline\]" $OUT/timeline_pipe$mode.err | tail -45 > $OUT/timeline_pipe$mode.txt; wc -l $OUT/timeline_pipe$mode.txt
done
