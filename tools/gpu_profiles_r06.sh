#!/bin/bash
# Collects the round-5 evidence files (copy gpurun_out/profiles_r06/* to profiles/):
#   pmc_r06_path.{txt,json}       counters over the WHOLE decode path (tools/gpu_pmc_path.sh: separate --pmc passes)
#   pmc_r06_traffic.json          HBM bytes per op label of the launch program + the kernel_source_hash of the build they were collected on
#                                 (bench.py's roofline.traffic quotes it for exactly these kernel sources only)
#   bench_r06.json                the default bench line (headline + other_configs + verify + alt_arith + cpu_baseline + measured ceilings)
#   per_op_r06.txt                per-op / per-level times of one DDIM iteration from a kernel trace (tools/trace_by_op.py), batch 32
#   per_op_r06_batch1.txt         the same at one image per call
#   rocprof_r06_kernel_stats.csv  rocprofv3 --kernel-trace --stats of `python bench.py` without the extra legs
#   parity_obs_r06.txt            every relative error the GPU test-suite observed (CDC_TEST_OBS), worst per test
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles_r06; mkdir -p $OUT
cd $R
timeout 1500 bash tools/gpu_pmc_path.sh 3 r06 > $OUT/pmc_path_log.txt 2>&1
cp $R/gpurun_out/pmc_r06/pmc_r06_path.txt $R/gpurun_out/pmc_r06/pmc_r06_path.json $R/gpurun_out/pmc_r06/pmc_r06_traffic.json $R/gpurun_out/pmc_r06/ops.txt $OUT/ 2>/dev/null
cp $R/gpurun_out/pmc_r06/pmc_r06_traffic.json $R/profiles/pmc_r06_traffic.json 2>/dev/null     # bench.py reads it (roofline.traffic)
tail -6 $OUT/pmc_path_log.txt
timeout 1200 python bench.py > $OUT/bench_r06.json 2> $OUT/bench_stderr.txt
tail -1 $OUT/bench_r06.json | cut -c1-300
bash tools/gpu_by_op.sh > $OUT/by_op_log.txt 2>&1
cp $R/gpurun_out/by_op/by_op_batch32.txt $OUT/per_op_r06.txt; cp $R/gpurun_out/by_op/by_op_batch1.txt $OUT/per_op_r06_batch1.txt
sed -i "s#$R/##" $OUT/per_op_r06.txt $OUT/per_op_r06_batch1.txt
( cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/rp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp -o k -- python $R/bench.py --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > $OUT/rocprof_bench_stdout.txt 2>&1 )
f=$(find $OUT/rp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_r06_kernel_stats.csv
rm -rf $OUT/rp
head -6 $OUT/rocprof_r06_kernel_stats.csv | cut -c1-160
rm -f $OUT/parity_obs.jsonl
CDC_TEST_OBS=$OUT/parity_obs.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
python3 - "$OUT" <<'PY'
import json, collections, sys, os
out = sys.argv[1]
obs = collections.defaultdict(list)
for l in open(os.path.join(out, "parity_obs.jsonl")):
    d = json.loads(l); obs[d["test"].split("::")[-1]].append(d["relerr"])
groups = collections.defaultdict(list)
for t, v in obs.items():
    groups[t.split("[")[0]].append((max(v), t))
with open(os.path.join(out, "parity_obs_r06.txt"), "w") as f:
    f.write("worst relative error (max|got - ref| / max(1, max|ref|)) observed per GPU parity test on the MI355X, all parametrisations\n")
    f.write("(tests/test_gpu_parity.py::relerr with CDC_TEST_OBS; the test bounds are about 3x these figures)\n")
    for base, lst in sorted(groups.items()):
        m = max(lst)
        f.write(f"{m[0]:.3e}  cases={len(lst):3d}  {base}   worst: {m[1][len(base):] or '-'}\n")
print(open(os.path.join(out, "parity_obs_r06.txt")).read()[-1200:])
PY
rm -f $OUT/parity_obs.jsonl
timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-alt-arith --no-extras --no-cpu-baseline > $OUT/bench_r06_driver_form.json 2>/dev/null
tail -1 $OUT/bench_r06_driver_form.json | cut -c1-200
