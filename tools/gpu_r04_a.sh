#!/bin/bash
# Round-4 first GPU call: the whole GPU test suite with every observed parity error logged (CDC_TEST_OBS), the default bench line
# with the per-op table, and the per-workgroup cycle categories of every conv_split2_kernel launch of one batch-32 iteration
# (development build libcdc_hip_timeline.so, tools/build_variant.sh timeline -DCDC_TIMELINE).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04a; mkdir -p $OUT; cd $R
rm -f $OUT/parity_obs.jsonl
CDC_TEST_OBS=$OUT/parity_obs.jsonl timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest.log
CDC_BENCH_OPS=400 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 600 $OUT/bench.json; grep "^\[op\]" $OUT/bench.err > $OUT/per_op.txt
python3 - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r04a/bench.json")).read().strip().splitlines()[-1])
print("value",d["value"],"ms/iter",d["roofline"]["ms_per_ddim_iter"],"frac",d["roofline"]["frac"],"verify",d.get("verify"))
for o in d.get("other_configs",[]): print(o["workload"],o["value"],o["ms_per_ddim_iter"],o["roofline"]["frac"],o["verify"])
print({k:round(v["ms_per_iteration"],3) for k,v in d["roofline"]["families"].items()})
PY
CDC_HIP_LIB=$R/cdc_compression_amd/libcdc_hip_timeline.so timeout 600 python bench.py --sample-steps 2 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > $OUT/timeline.json 2> $OUT/timeline.err
echo "timeline rc=$?"; grep -c "^\[timeline\]" $OUT/timeline.err
