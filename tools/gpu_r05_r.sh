#!/bin/bash
# Round 5: repeat the full-length batch-32 against batch-1 test (one run of the evidence collection saw 7.5e-5 on one row)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_r; mkdir -p $OUT; cd $R
for v in "CDC_X_UNUSED=1" "CDC_FOLD_TWO_LAUNCHES=1"; do
for i in 1 2 3 4 5 6 7 8; do
rm -f $OUT/obs.jsonl
env CDC_DEV=1 $v CDC_TEST_OBS=$OUT/obs.jsonl timeout 300 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k test_configs1_full_length 2>&1 | tail -1
python - <<PY
import json
print("[$v] run $i:", [round(json.loads(l)["relerr"]*1e6,2) for l in open("$OUT/obs.jsonl")])
PY
done; done
