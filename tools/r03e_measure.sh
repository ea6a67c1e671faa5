mkdir -p gpurun_out/r03e
python -m pytest tests/test_gpu_raster.py tests/test_gpu_fused_step.py tests/test_gpu_capacity.py tests/test_gpu_densify.py tests/test_gpu_variants.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider > gpurun_out/r03e/gputest.log 2>&1
tail -5 gpurun_out/r03e/gputest.log
python bench.py --steps 400 --warmup 50 --no-cpu-baseline > gpurun_out/r03e/bench_default.json 2> gpurun_out/r03e/bench_default.err
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render --densify-every 100 > gpurun_out/r03e/bench_densify100.json 2> gpurun_out/r03e/bench_densify100.err
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render --densify-every 100 --config 3 > gpurun_out/r03e/bench_densify100_c3.json 2> gpurun_out/r03e/bench_densify100_c3.err
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-ms-per-render --autograd > gpurun_out/r03e/bench_autograd.json 2> gpurun_out/r03e/bench_autograd.err
python - <<'PY'
import json
for f in ('default','densify100','densify100_c3','autograd'):
    try:
        d=json.load(open(f'gpurun_out/r03e/bench_{f}.json'))
        print(f, d['value'], d['ms_per_step'], d['ms_per_step_blocks']['median'], d['ms_per_step_blocks']['max'], d.get('densify'), d.get('ms_per_render_fwd_bwd'))
    except Exception as e:
        print(f, 'FAILED', e); print(open(f'gpurun_out/r03e/bench_{f}.err').read()[-2000:])
PY
python tools/prof_operator_path.py 2>&1 | head -8 > gpurun_out/r03e/prof_operator_bucket.txt
SKGS_COMPACT=1 python tools/prof_operator_path.py 2>&1 | head -8 > gpurun_out/r03e/prof_operator_compact.txt
cat gpurun_out/r03e/prof_operator_bucket.txt gpurun_out/r03e/prof_operator_compact.txt
