#!/bin/bash
# Round 5: the GPU suite four times over (one run of the evidence collection saw 7.5e-5 on one row of the full-length batch test)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_s; mkdir -p $OUT; cd $R
for i in 1 2 3 4; do
rm -f $OUT/obs$i.jsonl
CDC_TEST_OBS=$OUT/obs$i.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest$i.log 2>&1
tail -2 $OUT/pytest$i.log
python - <<PY
import json
print("run $i:", [(round(json.loads(l)["relerr"]*1e6,2)) for l in open("$OUT/obs$i.jsonl") if "configs1_full_length" in l])
PY
done
