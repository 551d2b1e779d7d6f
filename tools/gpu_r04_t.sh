#!/bin/bash
# Round-4: planes-only ResnetBlock-chain outputs + residual from planes -- U-Net tests first, then the per-op bench with (default) and
# without (CDC_NO_RESID_PF=1) the change
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04t; mkdir -p $OUT; cd $R
CDC_TEST_OBS=$OUT/obs.jsonl timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "unet or decode or taps or configs1 or batch32 or digest or compress or context or kodak" > $OUT/pytest_new.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest_new.log
python3 - <<PY
import json, collections
obs = collections.defaultdict(float)
for l in open('$OUT/obs.jsonl'):
    d = json.loads(l); k = d['test'].split('::')[-1].split('[')[0]; obs[k] = max(obs[k], d['relerr'])
print({k: float('%.3g' % v) for k, v in sorted(obs.items())})
PY
run_bench() {
    tag=$1; shift
    env "$@" CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-alt-arith --no-extras --no-other-configs > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    grep "^\[op\]" $OUT/bench_$tag.err > $OUT/per_op_$tag.txt
    python3 -c "
import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag ms/iter', round(d['roofline']['ms_per_ddim_iter'],3), 'verify', d.get('verify',{}).get('max_rel_err_vs_batch1_decode'), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()}, {k:round(v['ms_per_iteration'],3) for k,v in d['roofline']['families'].items()})"
}
run_bench new CDC_X=0
run_bench old CDC_DEV=1 ${OLD_ENV:-CDC_NO_RESID_PF=1}
grep -E "7x1| 128->192 | 192->256 " $OUT/per_op_new.txt $OUT/per_op_old.txt | grep -v "HOIST" | cut -c1-150 | head -12
