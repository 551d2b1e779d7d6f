#!/bin/bash
# Round-4: device-side timeline of conv_pf_kernel (variant `timeline`, -DCDC_TIMELINE): cycle categories per workgroup, dispatch ramp,
# workgroups per CU -- for every layer shape of one short decode.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04q; mkdir -p $OUT; cd $R
CDC_DEV=1 CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_timeline.so timeout 600 python bench.py --sample-steps 4 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs --no-verify > $OUT/bench.json 2> $OUT/bench.err
grep "^\[pf timeline\]" $OUT/bench.err | sort -u -k4,12 > $OUT/timeline.txt
wc -l $OUT/timeline.txt; cut -c1-700 $OUT/timeline.txt | head -40
