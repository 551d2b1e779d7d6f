#!/bin/bash
# Round-4 A/B of the software-pipelined tap loop of conv_split2_kernel (CDC_SPLIT2_PIPE = 0 off / 1 every fp16 shape / 2 NPW = 1 and
# stride-2 shapes): GPU tests in the default mode and with the loop forced everywhere, then the per-op table of a short bench per mode.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
rm -f $OUT/parity_obs.jsonl
CDC_TEST_OBS=$OUT/parity_obs.jsonl timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
CDC_SPLIT2_PIPE=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "conv2d or conv_transpose or unet_forward_matches_reference or decode_matches_reference or kodak_crops_500 or context_decoder or hyper_decoder or compressor_forward" > $OUT/pytest_pipe1.log 2>&1
echo "pytest(pipe=1) rc=$?"; tail -5 $OUT/pytest_pipe1.log
for mode in 0 1 2; do
  CDC_DEV=1 CDC_SPLIT2_PIPE=$mode CDC_BENCH_OPS=400 timeout 600 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  grep "^\[op\]" $OUT/bench_$mode.err > $OUT/per_op_$mode.txt
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$mode.json').read().strip().splitlines()[-1])
print('mode $mode: value %.3f  ms/iter %.3f' % (d['value'], d['roofline']['ms_per_ddim_iter']), {k:round(v['ms_per_iteration'],3) for k,v in d['roofline']['families'].items()})"
done
