#!/bin/bash
# Round 5: conv_ws_kernel iteration: its tests + the per-op times of the trunk layers in the bench
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_c; mkdir -p $OUT; cd $R
export CDC_DEV=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "weight_stationary" > $OUT/pytest_ws.log 2>&1
tail -3 $OUT/pytest_ws.log
CDC_BENCH_OPS=400 timeout 300 python bench.py --sample-steps 100 --prof-every 5 --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/bench.json 2> $OUT/err.txt
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('ms/iter', d['roofline']['ms_per_ddim_iter'], 'verify', d.get('verify'))"
grep -E " WS|ln C=" $OUT/err.txt | sort -k8 | head -40
