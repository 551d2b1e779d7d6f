#!/bin/bash
# Round 5, first call: the GPU test-suite on the round's first build + the headline bench line with the per-op table
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_a; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
CDC_BENCH_OPS=400 timeout 900 python bench.py --no-other-configs --no-alt-arith --no-extras > $OUT/bench.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op.txt
tail -1 $OUT/bench.json | cut -c1-400
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r05_a/bench.json")).read().strip().splitlines()[-1])
r=d["roofline"]
print("ms/iter", r["ms_per_ddim_iter"], "frac", r["frac"], "mfma", r["mfma_sustained_measured"], "hbm", r["hbm_copy_measured"], "batch1", d.get("batch1"))
PY
