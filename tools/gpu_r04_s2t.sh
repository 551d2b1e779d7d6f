#!/bin/bash
# Development aid (round 4, second session): MFMA issue order (snake over the (m, n) block, product groups meeting in a shared operand)
# against the row-major order: two builds of the library on one box, alternating (libcdc_hip_old.so = the build before the change).
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2t; mkdir -p $O
F="--steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
for v in old new old new old new; do
  if [ $v = old ]; then export CDC_HIP_LIB=$GRAFT_REPO_ROOT/cdc_compression_amd/libcdc_hip_old.so; else unset CDC_HIP_LIB; fi
  timeout 300 python bench.py $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$v', 'ms/iter', round(j['ms_per_step']/100,4), 'pf3', round(r['families']['conv_pf3_kernel']['ms_per_iteration'],3), 'pf', round(r['families']['conv_pf_kernel']['ms_per_iteration'],3), 'dominant', round(r['avg_launch_ms'],4), 'pair2', round(r['by_shape'][1]['avg_launch_ms'],4))
"
done 2>&1 | tee $O/ab.txt
unset CDC_HIP_LIB
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "persistent_ping_pong or configs1 or full_resolution or batch32 or unet_forward_matches" > $O/pytest.log 2>&1
tail -2 $O/pytest.log
