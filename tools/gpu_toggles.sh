#!/bin/bash
# Development aid: whole-model A/B of planner / builder switches on ONE box (ms per 20-iteration decode).
run() { echo -n "$1: "; env $2 python bench.py --steps 2 --warmup 1 --sample-steps 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run base ""
for t in "$@"; do run "$t" "$t=1"; done
run base ""
