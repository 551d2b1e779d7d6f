#!/bin/bash
# Development aid: per-kernel average durations (rocprofv3 --kernel-trace --stats) of one command.
# usage: gpu_kstats.sh <tag> <command...>
set -u
TAG="$1"; shift
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- "$@" > $OUT/stdout.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
