#!/bin/bash
# Development aid (round 4, second session): batch 1 -- more K slices (CDC_KMAX=8, CDC_KS_TARGET) on the split-K + LayerNorm layers:
# per-class sums of the per-op table (convolutions by level, LayerNorm passes) and ms per iteration.
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2l; mkdir -p $O
F="--batch 1 --steps 1 --warmup 1 --sample-steps 100 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs"
for cfg in "X=0" "CDC_KMAX=8" "CDC_KMAX=8 CDC_KS_TARGET=2048" "CDC_KMAX=6"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env CDC_DEV=1 $cfg CDC_BENCH_OPS=400 timeout 300 python bench.py $F 2>&1 >/dev/null | grep "^\[op\]" > $O/ops_$tag.txt
  echo -n "$cfg (events on): "; python3 - $O/ops_$tag.txt <<'PY'
import re,sys,collections
c=collections.defaultdict(float)
for l in open(sys.argv[1]):
    m=re.match(r'\[op\]\s+([\d.]+) ms\s+[\d.]+ TF\s+(.*)',l); ms=float(m.group(1)); d=m.group(2)
    if d.startswith('conv'):
        o=re.search(r'out\s+(\d+)x',d).group(1); k='conv@'+o+(' ks' if re.search(r'ks[2-9]',d) else '')
    else: k=d.split()[0]
    c[k]+=ms
print(round(sum(c.values()),3), {k:round(v,3) for k,v in sorted(c.items())})
PY
  echo -n "$cfg: "; env CDC_DEV=1 $cfg timeout 300 python bench.py $F 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']/100,4), 'ms/iter')"
done 2>&1 | tee $O/kmax_b1.txt
