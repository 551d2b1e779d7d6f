#!/bin/bash
# Round 5: the GPU suite + the bench line (per-op table) of the pruned build
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_h; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
CDC_BENCH_OPS=400 timeout 900 python bench.py --no-other-configs --no-alt-arith --no-extras > $OUT/bench.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op.txt
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r05_h/bench.json")).read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "ms/iter", r["ms_per_ddim_iter"], "frac", r["frac"], "verify", d.get("verify"), "batch1", d.get("batch1"))
PY
