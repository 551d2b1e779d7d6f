#!/bin/bash
# Development aid (round 4, second session): HEAD sanity -- full GPU suite, smoke, one bench line with per-op table.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2a
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s2a/pytest.log 2>&1
tail -4 gpurun_out/s2a/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
CDC_BENCH_OPS=200 timeout 900 python bench.py --no-cpu-baseline --no-alt-arith --no-other-configs > gpurun_out/s2a/bench.json 2> gpurun_out/s2a/bench_stderr.txt
grep "^\[op\]" gpurun_out/s2a/bench_stderr.txt > gpurun_out/s2a/per_op.txt
cut -c1-400 gpurun_out/s2a/bench.json
