#!/bin/bash
# Round 5: per-step determinism stress on ONE box: the counted wait of round 4 first (a box that shows nothing in 35 repetitions is not
# worth the longer run: the event rate differs between boxes), then the default (vmcnt 0 per step) for longer
set -u
R=$GRAFT_REPO_ROOT; cd $R
echo "== CDC_PW_COUNTED_WAIT=1"
env CDC_DEV=1 CDC_PW_COUNTED_WAIT=1 timeout 1500 python tools/determinism_stress_steps.py 35 2>&1 | grep -v amdgpu.ids | grep -v "^taps" | cut -c1-330 | tail -6 | tee /tmp/probe.txt
if grep -q " 0 differ" /tmp/probe.txt; then echo "no event with the counted wait on this box: stop"; exit 0; fi
echo "== default (vmcnt 0 per step)"
env CDC_DEV=1 timeout 1500 python tools/determinism_stress_steps.py 90 2>&1 | grep -v amdgpu.ids | grep -v "^taps" | cut -c1-330 | tail -6
