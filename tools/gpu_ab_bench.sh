#!/bin/bash
# usage: ab_bench.sh  -> runs bench with old & new libs, prints value and class ms
cd $GRAFT_REPO_ROOT
for v in old new old new; do
  if [ $v = old ]; then export CDC_HIP_LIB=$GRAFT_REPO_ROOT/cdc_compression_amd/libcdc_hip_old.so; else unset CDC_HIP_LIB; fi
  python bench.py --steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$v', round(j['value']*j['config']['sample_steps']/500,4), 'img/s-eq', 'ms/iter', round(r['ms_per_ddim_iter'],3), {k:round(v,3) for k,v in r['class_ms_per_ddim_iter'].items()})
"
done
