#!/bin/bash
# Development aid: build an A/B variant of the library with extra compiler flags.
# usage: build_variant.sh NAME "-DSOME_SWITCH ..."   ->  cdc_compression_amd/libcdc_hip_NAME.so
# Select it at run time with CDC_HIP_LIB=<path> (see cdc_compression_amd/_lib.py).
set -eu
NAME="$1"; EXTRA="${2:-}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="${SRC_DIR:-$ROOT/cdc_compression_amd/csrc}"; OBJ="/tmp/cdc_variant_$NAME"; mkdir -p "$OBJ"
pids=()
for f in "$SRC"/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$SRC" -I"$SRC/../../include" -Wno-unused-result $EXTRA -c "$f" -o "$OBJ/$(basename "$f" .hip).o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/cdc_compression_amd/libcdc_hip_$NAME.so" "$OBJ"/*.o
echo "built libcdc_hip_$NAME.so"
