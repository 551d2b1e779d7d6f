#!/bin/bash
# Development aid: time one conv shape under several library variants.  usage: gpu_variants.sh "shape" v1 v2 ...
S="$1"; shift
echo -n "base      : "; TUNE_CHILD=1 python tools/gpu_conv_tune.py $S 2>/dev/null | tail -1
for v in "$@"; do
  printf "%-10s: " $v; CDC_HIP_LIB=$PWD/cdc_compression_amd/libcdc_hip_$v.so TUNE_CHILD=1 python tools/gpu_conv_tune.py $S 2>/dev/null | tail -1
done
