#!/bin/bash
# Round 5: the weight-stationary trunk kernel -- its own tests first, then the suite, then the bench line with the per-op table
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_b; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "weight_stationary or planes_only or probes" > $OUT/pytest_ws.log 2>&1
tail -15 $OUT/pytest_ws.log
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
CDC_BENCH_OPS=400 timeout 900 python bench.py --no-other-configs --no-alt-arith --no-extras > $OUT/bench.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op.txt
tail -3 $OUT/bench_stderr.txt
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r05_b/bench.json")).read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "ms/iter", r["ms_per_ddim_iter"], "frac", r["frac"], "verify", d.get("verify"), "batch1", d.get("batch1"))
print("mfma", r["mfma_sustained_measured"]["tflops_random_operands"], r["mfma_sustained_measured"]["tflops_constant_operands"], "hbm", r["hbm_copy_measured"]["gb_per_s"])
PY
grep -E " WS|ln C=" $OUT/per_op.txt | head -60
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/wino_lab tools/ubench/wino_lab.hip 2>/dev/null && true # > $OUT/wino_lab.txt 2>&1
cat $OUT/wino_lab.txt
