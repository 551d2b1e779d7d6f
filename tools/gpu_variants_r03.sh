cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/variants_r03
timeout 600 python bench.py --size 512 --batch 16 --no-cpu-baseline --no-extras > gpurun_out/variants_r03/bench_r03_512_b16.json 2> gpurun_out/variants_r03/e1.txt
timeout 600 python bench.py --param eps --no-cpu-baseline --no-extras > gpurun_out/variants_r03/bench_r03_eps_1000step.json 2> gpurun_out/variants_r03/e2.txt
timeout 600 python bench.py --batch 1 --no-cpu-baseline --no-extras --no-alt-arith > gpurun_out/variants_r03/bench_r03_batch1.json 2> gpurun_out/variants_r03/e3.txt
for f in gpurun_out/variants_r03/*.json; do python3 -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); r=j['roofline']
print('$f', round(j['value'],4), j['unit'], 'ms/iter', round(r['ms_per_ddim_iter'],3), r['launch_key'], round(r['frac'],3), j.get('verify'), j.get('range_guard'))
"; done
