#!/bin/bash
# Counter-based HBM traffic + MFMA / LDS counters of the dominant launch shape of bench.py's default workload
# (block2 of the 64-channel ResnetBlocks: 64->64 3x3 on pre-split planes + fused LN + residual @256x256, batch 32:
# conv_pf_kernel; run with CDC_PF=1 TUNE_RESID=1 so that the tune tool launches that kernel), separate --pmc passes (no tracing domain besides --kernel-trace),
# per /opt/skills/guides/MI355X_MICROARCH.md: hbm_bytes = FETCH_SIZE*1024*2 (gfx950 reports 1/2 of a wide coalesced
# read) + WRITE_SIZE*1024.  Writes profiles-ready files under gpurun_out/pmc_r02/ (copy them to profiles/).
set -u
export CDC_DEV=1      # the CDC_* planner switches below are development switches (cdc_internal.h: dev_env)
SHAPE="${1:-32 64 256 256 64 3 1 1}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r02
mkdir -p $OUT
bash $GRAFT_REPO_ROOT/tools/gpu_pmc_conv.sh "$SHAPE" > $OUT/pmc_r02_conv3x3_mfma.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/gpu_pmc_traffic.sh "$SHAPE" > $OUT/pmc_r02_conv3x3_traffic.txt 2>&1
python3 - "$OUT" "$SHAPE" <<'PY'
import json, re, sys, os
out, shape = sys.argv[1], [int(v) for v in sys.argv[2].split()]
B, Ci, H, W, Co, k, s, ln = shape
txt = open(os.path.join(out, "pmc_r02_conv3x3_traffic.txt")).read()
m = re.search(r"= ([0-9.]+) MB per launch", txt)
arith = 0 if os.environ.get("CDC_ARITH") == "0" else 1
if m:
    kern = "PF" if os.environ.get("CDC_PF") == "1" else ("SPLIT2H" if arith else "SPLIT2")
    res = " +res" if os.environ.get("TUNE_RESID") else ""
    json.dump({"launch": f"B{B} conv {k}x{k} s{s} {Ci}->{Co} out {H//s}x{W//s} MB2 NPW2 {kern}{res}", "arith": arith,
               "hbm_bytes_per_launch": float(m.group(1)) * 1e6,
               "source": "profiles/pmc_r02_conv3x3_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                         "FETCH_SIZE x2 gfx950 correction)"}, open(os.path.join(out, "pmc_r02_traffic.json"), "w"))
print(txt)
PY
cat $OUT/pmc_r02_conv3x3_mfma.txt
