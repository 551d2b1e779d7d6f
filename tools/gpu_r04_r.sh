#!/bin/bash
# Round-4: conv_pf_kernel tests + timeline (variant `timeline`) + per-op bench, after a change to its DMA issue
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04r; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "pre_split or stride2 or fused_phases or stage_taps or alternate_kernel_modes or pointwise or persistent" > $OUT/pytest_new.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_new.log
CDC_DEV=1 CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_timeline.so timeout 600 python bench.py --sample-steps 4 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs --no-verify > $OUT/bench_tl.json 2> $OUT/bench_tl.err
grep "^\[pf timeline\]" $OUT/bench_tl.err | sort -u -k4,12 | sed 's/: [0-9]* workgroups.*per workgroup (wave 0):/:/' > $OUT/timeline.txt
cat $OUT/timeline.txt | cut -c1-300
CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench.json 2> $OUT/bench.err
grep "^\[op\]" $OUT/bench.err > $OUT/per_op.txt
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), 'verify', d.get('verify',{}).get('max_rel_err_vs_batch1_decode'), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()}, {k:round(v['ms_per_iteration'],3) for k,v in d['roofline']['families'].items()})"
grep -E " s2 | TZ4|1x7" $OUT/per_op.txt | head -8
