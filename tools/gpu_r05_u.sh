#!/bin/bash
# Round 5: per-step determinism stress (which launch of the 64x64 attention level differs between runs?)
set -u
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python tools/determinism_stress_steps.py ${1:-60} 2>&1 | grep -v amdgpu.ids | grep -v "^taps" | tail -12
