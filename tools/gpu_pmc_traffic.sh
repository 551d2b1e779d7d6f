#!/bin/bash
# HBM traffic of one convolution shape from PMC counters (separate passes, no tracing domains besides
# --kernel-trace), per MI355X_MICROARCH.md: hbm_bytes = FETCH_SIZE*1024*2 (gfx950 reports 1/2 of a wide
# coalesced read) + WRITE_SIZE*1024 (uncalibrated).   usage: gpu_pmc_traffic.sh "<B Cin H W Cout k s LN>"
set -u
export CDC_DEV=1      # the CDC_* planner switches below are development switches (cdc_internal.h: dev_env)
SHAPE="$1"
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  TUNE_CHILD=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python $GRAFT_REPO_ROOT/tools/gpu_conv_tune.py $SHAPE > /dev/null 2>&1
done
python3 - "$OUT" "$SHAPE" <<'PY'
import csv, glob, sys, collections
out, shape = sys.argv[1], [int(v) for v in sys.argv[2].split()]
B, Ci, H, W, Co, k, s, ln = shape
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "conv_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:48]].append(float(r["Counter_Value"]))
    for kname, v in acc.items():
        vals.setdefault(kname, {})[c] = sum(v) / len(v)
Ho, Wo = H // s, W // s
import os
res = bool(os.environ.get("TUNE_RESID"))
alg = 4.0 * (B * Ci * H * W + B * Co * Ho * Wo * (2 if res else 1))
for kname, v in vals.items():
    fetch = v.get("FETCH_SIZE", 0) * 1024 * 2
    write = v.get("WRITE_SIZE", 0) * 1024
    print(f"{kname}: FETCH_SIZE(KB)={v.get('FETCH_SIZE',0):.0f} WRITE_SIZE(KB)={v.get('WRITE_SIZE',0):.0f} "
          f"-> hbm read {fetch/1e6:.1f} MB (x2 corrected) + write {write/1e6:.1f} MB = {(fetch+write)/1e6:.1f} MB per launch; "
          f"algorithmic {alg/1e6:.1f} MB (input once + output once" + (" + residual once" if res else "") + ")")
PY
