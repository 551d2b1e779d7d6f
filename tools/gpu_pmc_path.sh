#!/bin/bash
# Round-3 evidence: counters over the WHOLE decode path, not one launch.  Runs bench.py's default workload with a short
# DDIM schedule under rocprofv3 in separate passes (the pool refuses --pmc together with tracing domains other than
# --kernel-trace; /opt/skills/guides/MI355X_MICROARCH.md prescribes separate passes):
#   pass t: --kernel-trace --stats                       -> per-kernel time
#   pass a / w: --pmc FETCH_SIZE, --pmc WRITE_SIZE (the two do not fit one pass) -> HBM bytes per dispatch (FETCH_SIZE x2: gfx950 correction of the guide)
#   pass b: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES
#   pass c: --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM
# and tools/pmc_path_summary.py turns them into profiles-ready text / json under gpurun_out/pmc_<tag>/
# (pmc_<tag>_path.{txt,json}: per kernel; pmc_<tag>_traffic.json: per op label of the launch program, read by bench.py).
# usage (on the GPU box): bash tools/gpu_pmc_path.sh [sample_steps=4] [round tag=r04]
set -u
STEPS="${1:-4}"
TAG="${2:-r05}"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --sample-steps $STEPS --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs --prof-every 1000000 --dump-ops $OUT/ops.txt"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o p -- $CMD > $OUT/bench_t.json 2> $OUT/bench_t.err
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/a -o p -- $CMD > /dev/null 2> $OUT/bench_a.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- $CMD > /dev/null 2> $OUT/bench_w.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o p -- $CMD > /dev/null 2> $OUT/bench_b.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/c -o p -- $CMD > /dev/null 2> $OUT/bench_c.err
python3 $ROOT/tools/pmc_path_summary.py $OUT $STEPS $TAG $OUT/ops.txt > $OUT/pmc_${TAG}_path.txt 2> $OUT/summary.err
cat $OUT/pmc_${TAG}_path.txt; cat $OUT/summary.err
# keep the merge small: the raw per-dispatch tables are not needed once summarised
find $OUT -name "*counter_collection.csv" -size +4M -delete
find $OUT -name "*kernel_trace.csv" -size +4M -delete
