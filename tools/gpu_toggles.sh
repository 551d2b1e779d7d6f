#!/bin/bash
# Development aid: whole-model A/B of planner / builder switches on ONE box (ms per DDIM iteration + per-class ms).
# usage: gpu_toggles.sh "VAR=val [VAR2=val2]" ...      (each argument is one configuration; CDC_DEV=1 is implied)
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env CDC_DEV=1 $2 python bench.py --steps 1 --warmup 1 --sample-steps 60 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --prof-every 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print(round(r['ms_per_ddim_iter'],3), 'ms/iter', {k:round(v,3) for k,v in r['class_ms_per_ddim_iter'].items()})"; }
run base ""
for t in "$@"; do run "$t" "$t"; done
run base ""
