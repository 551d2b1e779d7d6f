#!/bin/bash
# Development aid: PMC counters (separate passes) for one convolution shape via tools/gpu_conv_tune.py child.
# usage: gpu_pmc_conv.sh "<B Cin H W Cout k s LN>" [CDC_PLAN]
set -u
export CDC_DEV=1      # the CDC_* planner switches below are development switches (cdc_internal.h: dev_env)
SHAPE="$1"; PLAN="${2:-}"
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  TUNE_CHILD=1 CDC_PLAN="$PLAN" rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/gpu_conv_tune.py $SHAPE > /dev/null 2>&1
  python3 - "$OUT/$name" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    if "conv_" in k:
        print(k, {c: round(x / max(cnt[(k, c)], 1)) for c, x in v.items()})
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS
run c SQ_WAVES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_CYCLES
