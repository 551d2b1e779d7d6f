#!/bin/bash
# Round 5: fused-phase Upsample epilogue without scratch spills; conv_ws1_kernel from 4 workgroups
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/suite_bench; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
CDC_BENCH_OPS=400 timeout 900 python bench.py --no-other-configs --no-alt-arith --no-extras > $OUT/bench.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op.txt
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/suite_bench/bench.json")).read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "ms/iter", r["ms_per_ddim_iter"], "frac", r["frac"], "verify", d.get("verify"), "batch1", d.get("batch1"))
PY
for b in 1 2 4 8; do
timeout 300 python bench.py --batch $b --sample-steps 60 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b: ms/iter', d['roofline']['ms_per_ddim_iter'])"
done
for g in 0 1; do
CDC_GRAPH=$g timeout 300 python bench.py --batch 1 --sample-steps 100 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CDC_GRAPH=$g batch 1: ms/iter', d['roofline']['ms_per_ddim_iter'])"
done
