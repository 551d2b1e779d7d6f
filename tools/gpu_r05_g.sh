#!/bin/bash
# Round 5: conv_ws_kernel without LayerNorm on load, 64- / 128-pixel tiles: its tests, the GPU suite, the bench line with the per-op table
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_g; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "weight_stationary or few_pixel or linear_attention" > $OUT/pytest_ws.log 2>&1
tail -4 $OUT/pytest_ws.log
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
CDC_BENCH_OPS=400 timeout 900 python bench.py --no-other-configs --no-alt-arith --no-extras > $OUT/bench.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op.txt
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r05_g/bench.json")).read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "ms/iter", r["ms_per_ddim_iter"], "frac", r["frac"], "verify", d.get("verify"), "batch1", d.get("batch1"))
PY
CDC_DEV=1 CDC_WS_MIN_WGS=100000 timeout 300 python bench.py --batch 1 --sample-steps 100 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 1 without conv_ws_kernel: ms/iter', d['roofline']['ms_per_ddim_iter'])"
timeout 300 python bench.py --batch 1 --sample-steps 100 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 1 with conv_ws_kernel: ms/iter', d['roofline']['ms_per_ddim_iter'])"
for b in 2 4 8; do
CDC_DEV=1 CDC_WS_MIN_WGS=100000 timeout 300 python bench.py --batch $b --sample-steps 60 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b without: ms/iter', d['roofline']['ms_per_ddim_iter'])"
timeout 300 python bench.py --batch $b --sample-steps 60 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b with: ms/iter', d['roofline']['ms_per_ddim_iter'])"
done
