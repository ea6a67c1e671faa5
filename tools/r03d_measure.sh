mkdir -p gpurun_out/r03d
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render --densify-every 100 > gpurun_out/r03d/bench_densify100.json 2> gpurun_out/r03d/bench_densify100.err
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render --densify-every 100 --config 3 > gpurun_out/r03d/bench_densify100_c3.json 2> gpurun_out/r03d/bench_densify100_c3.err
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render --config 3 > gpurun_out/r03d/bench_steady_c3.json 2> gpurun_out/r03d/bench_steady_c3.err
python - <<'PY'
import json
for f in ('densify100','densify100_c3','steady_c3'):
    try:
        d=json.load(open(f'gpurun_out/r03d/bench_{f}.json'))
        print(f, d['value'], d['ms_per_step'], d['ms_per_step_blocks']['median'], d['ms_per_step_blocks']['max'], d.get('densify'))
    except Exception as e:
        print(f, 'FAILED', e); print(open(f'gpurun_out/r03d/bench_{f}.err').read()[-2000:])
PY
python tools/prof_operator_path.py > gpurun_out/r03d/prof_operator_path.txt 2>&1
head -45 gpurun_out/r03d/prof_operator_path.txt
