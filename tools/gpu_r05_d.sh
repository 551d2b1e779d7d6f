#!/bin/bash
# Round 5: device-side timeline of conv_ws_kernel (-DCDC_WS_LAB build, s_memtime stamps of wave 0 of every workgroup) + tests + per-op
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_d; mkdir -p $OUT; cd $R
export CDC_DEV=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "weight_stationary" > $OUT/pytest_ws.log 2>&1
tail -3 $OUT/pytest_ws.log
CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_wslab.so CDC_WS_TL=1 CDC_NO_RANGE_GUARD=1 timeout 300 python bench.py --sample-steps 30 --prof-every 2 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/bench_tl.json 2> $OUT/err_tl.txt
grep "ws-tl" $OUT/err_tl.txt | sed 's/span [0-9]* | mean since start://' | cut -c1-300
for v in 0 1; do
[ $v = 1 ] && export CDC_NO_WS_LNLOAD=1
CDC_BENCH_OPS=400 timeout 300 python bench.py --sample-steps 100 --prof-every 5 --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/bench_$v.json 2> $OUT/err_$v.txt
python -c "
import json
d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
print('NO_LNLOAD=$v ms/iter', d['roofline']['ms_per_ddim_iter'], 'verify', d.get('verify'))"
grep -E " WS|ln C=" $OUT/err_$v.txt | sort -k8 | awk '{s+=$2} END {print "sum of WS + ln ops:", s}'
grep -E " WS" $OUT/err_$v.txt | sort -k8 | head -30
done
