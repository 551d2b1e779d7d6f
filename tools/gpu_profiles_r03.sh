#!/bin/bash
# Collects the round-3 evidence files (copy gpurun_out/profiles_r03/* to profiles/):
#   bench_r03.json               the default bench line
#   per_op_r03.txt               per-op hipEvent averages of the same run (CDC_BENCH_OPS)
#   rocprof_r03_kernel_stats.csv rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --no-extras`
#   pmc_r03_path.{txt,json}      counters over the WHOLE decode path (tools/gpu_pmc_path.sh)
#   pmc_r03_conv3x3_*.txt, pmc_r03_traffic.json   PMC passes of the dominant launch shape alone (128->128 3x3 @128x128 + residual on
#                                conv_pf3_kernel; tools/gpu_pmc_conv.sh / gpu_pmc_traffic.sh through the single-op entry point)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles_r03; mkdir -p $OUT
cd $R
# whole-path counters first: bench.py reads profiles/pmc_r03_path.json (roofline.traffic)
timeout 1500 bash tools/gpu_pmc_path.sh 3 > $OUT/pmc_path_log.txt 2>&1
cp $R/gpurun_out/pmc_r03/pmc_r03_path.txt $R/gpurun_out/pmc_r03/pmc_r03_path.json $OUT/ 2>/dev/null
cp $R/gpurun_out/pmc_r03/pmc_r03_path.json $R/profiles/pmc_r03_path.json 2>/dev/null
( export CDC_DEV=1 CDC_PF=1 TUNE_RESID=1
: > $OUT/pmc_r03_conv3x3_mfma.txt; : > $OUT/pmc_r03_conv3x3_traffic.txt
for SHAPE in "32 64 256 256 64 3 1 1" "32 128 128 128 128 3 1 1"; do
  echo "== shape B Cin H W Cout k s LN = $SHAPE (fused LayerNorm + residual operand, fp32 output)" | tee -a $OUT/pmc_r03_conv3x3_mfma.txt >> $OUT/pmc_r03_conv3x3_traffic.txt
  timeout 600 bash tools/gpu_pmc_conv.sh "$SHAPE" >> $OUT/pmc_r03_conv3x3_mfma.txt 2>&1
  timeout 600 bash tools/gpu_pmc_traffic.sh "$SHAPE" >> $OUT/pmc_r03_conv3x3_traffic.txt 2>&1
done
python3 - "$OUT" <<'PY'
import json, re, sys, os
out = sys.argv[1]
txt = open(os.path.join(out, "pmc_r03_conv3x3_traffic.txt")).read()
res = {}
for sec in txt.split("== shape")[1:]:
    B, Ci, H, W, Co = [int(v) for v in sec.split("=")[1].split("(")[0].split()[:5]]
    m = None
    for m in re.finditer(r"conv_pf3\S*.*?= ([0-9.]+) MB per launch; algorithmic ([0-9.]+) MB", sec):
        pass
    if m:
        res[f"B{B} conv 3x3 s1 {Ci}->{Co} out {H}x{W} PF3"] = {
            "arith": 1, "hbm_bytes_per_launch": float(m.group(1)) * 1e6, "algorithmic_bytes_of_this_variant": float(m.group(2)) * 1e6,
            "variant": "fused LayerNorm + residual operand, fp32 output (one of the epilogue variants of the pair)",
            "source": "profiles/pmc_r03_conv3x3_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 gfx950 "
                      "correction), the layer launched alone through cdc_op_conv2d"}
json.dump(res, open(os.path.join(out, "pmc_r03_traffic.json"), "w"), indent=1)
print(txt[-1800:])
PY
tail -5 $OUT/pmc_r03_conv3x3_mfma.txt | cut -c1-600
)
cp $OUT/pmc_r03_traffic.json $R/profiles/pmc_r03_traffic.json 2>/dev/null
CDC_BENCH_OPS=400 timeout 900 python bench.py > $OUT/bench_r03.json 2> $OUT/bench_stderr.txt
grep "^\[op\]" $OUT/bench_stderr.txt > $OUT/per_op_r03.txt
tail -1 $OUT/bench_r03.json | cut -c1-400
( cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/rp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp -o k -- python $R/bench.py --no-cpu-baseline --no-verify --no-alt-arith --no-extras > $OUT/rocprof_bench_stdout.txt 2>&1 )
f=$(find $OUT/rp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_r03_kernel_stats.csv
rm -rf $OUT/rp
head -8 $OUT/rocprof_r03_kernel_stats.csv | cut -c1-200
