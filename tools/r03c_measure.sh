mkdir -p gpurun_out/r03c
(time python -m pytest tests -q -m gpu -p no:cacheprovider) > gpurun_out/r03c/gputest.log 2>&1
tail -8 gpurun_out/r03c/gputest.log
cp gpurun_out/parity_observed.json gpurun_out/r03c/parity_observed_full.json
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render > gpurun_out/r03c/bench_steady.json 2> gpurun_out/r03c/bench_steady.err
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render --densify-every 100 > gpurun_out/r03c/bench_densify100.json 2> gpurun_out/r03c/bench_densify100.err
python - <<'PY'
import json
for f in ('steady','densify100'):
    try:
        d=json.load(open(f'gpurun_out/r03c/bench_{f}.json'))
        print(f, d['value'], d['ms_per_step'], d.get('densify'))
    except Exception as e:
        print(f, 'FAILED', e); print(open(f'gpurun_out/r03c/bench_{f}.err').read()[-2000:])
PY
python examples/train_views.py --iters 450 --densify-every 100 --gaussians 50000 --size 400 > gpurun_out/r03c/example_capacity.txt 2>&1
python examples/train_views.py --iters 450 --densify-every 100 --gaussians 50000 --size 400 --capacity 0 > gpurun_out/r03c/example_rebuild.txt 2>&1
grep -h "densify\|it/s\|iters" gpurun_out/r03c/example_capacity.txt | tail -8
grep -h "densify\|it/s\|iters" gpurun_out/r03c/example_rebuild.txt | tail -8
