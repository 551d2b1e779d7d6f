#!/bin/bash
# Development aid: A/B time conv shapes with two builds of the library.  usage: gpu_ab.sh VARIANT "shape" ["shape" ...]
V="$1"; shift
for s in "$@"; do
  for rep in 1 2; do
    echo -n "base  : "; TUNE_CHILD=1 python tools/gpu_conv_tune.py $s | tail -1
    echo -n "$V : "; CDC_HIP_LIB=$PWD/cdc_compression_amd/libcdc_hip_$V.so TUNE_CHILD=1 python tools/gpu_conv_tune.py $s | tail -1
  done
done
