cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04j
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "attention or unet_forward or decode_matches or kodak_crops_500 or full_resolution or x_param_512" > gpurun_out/r04j/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04j/pytest.log
for m in 0 1; do
  env=""; [ $m = 0 ] && env="CDC_DEV=1 CDC_NO_FOLD_MFMA=1"
  env $env CDC_BENCH_OPS=400 python bench.py --sample-steps 100 --prof-every 10 --no-cpu-baseline --no-verify --no-alt-arith --no-extras --no-other-configs > gpurun_out/r04j/bench_$m.json 2> gpurun_out/r04j/bench_$m.err
  grep "^\[op\]" gpurun_out/r04j/bench_$m.err | grep "ctxf" | tr '\n' ';'; echo
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04j/bench_$m.json').read().strip().splitlines()[-1]); print('mode $m', round(d['roofline']['ms_per_ddim_iter'],3), {k:round(v,3) for k,v in d['roofline']['class_ms_per_ddim_iter'].items()})"
done
