#!/bin/bash
# Development aid (round 4, second session): attention fold R1 / R2 with LDS-DMA staging, straight-line K loops and the bias operand in LDS --
# parity of the attention / U-Net tests, per-op ctxf times at batch 32 and 1, whole-model ms per iteration (CDC_FOLD_RT=1: run-time channel count).
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "attention or unet_forward or decode_matches or compressor" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
for B in 32 1; do
  CDC_BENCH_OPS=400 timeout 300 python bench.py --batch $B $F 2>&1 >/dev/null | grep "^\[op\].*ctxf" > $O/ctxf_b$B.txt
  CDC_DEV=1 CDC_FOLD_RT=1 CDC_BENCH_OPS=400 timeout 300 python bench.py --batch $B $F 2>&1 >/dev/null | grep "^\[op\].*ctxf" > $O/ctxf_rt_b$B.txt
  echo "batch $B ctxf (compile-time count | run-time count):"; paste <(awk '{print $2}' $O/ctxf_b$B.txt) <(awk '{print $2}' $O/ctxf_rt_b$B.txt) <(cut -c30- $O/ctxf_b$B.txt)
done
run() { echo -n "batch $2 $1: "; env CDC_DEV=1 $1 timeout 300 python bench.py --batch $2 $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']/100,4), 'ms/iter')"; }
{ for B in 1 32; do run "X=0" $B; run "CDC_FOLD_RT=1" $B; run "X=0" $B; done; } 2>&1 | tee $O/ab.txt
