#!/bin/bash
# Round 5 experiment: co-resident workgroups of conv_pf_kernel out of phase (PfArgs::stagger)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_k; mkdir -p $OUT; cd $R
for s in 0 1 2 3 4 6; do
CDC_PF_STAGGER=$s CDC_BENCH_OPS=400 timeout 300 python bench.py --sample-steps 100 --prof-every 5 --no-verify --no-cpu-baseline --no-other-configs --no-alt-arith --no-extras > $OUT/bench_$s.json 2> $OUT/err_$s.txt
echo "== stagger $s: ms/iter $(python -c "import json;print(json.loads(open('$OUT/bench_$s.json').read().strip().splitlines()[-1])['roofline']['ms_per_ddim_iter'])")"
grep " PF " $OUT/err_$s.txt | grep -E "TZ4|s2 |7x1|1x7|192->192  out  64|256->256" | awk '{printf "%s ", $2; for(i=5;i<=13;i++) printf "%s ", $i; print $NF}' | sort -k4 | head -14
done
