bash tools/gpu_tests.sh final 1; echo "GPU TESTS rc=$?"
python bench.py > gpurun_out/final_default.json 2> gpurun_out/final_default.err; cut -c1-250 gpurun_out/final_default.json
SKGS_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist1', d['value'], d['ms_per_step'])"
for c in 0 2 3 4; do python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c', d['value'], d['ms_per_step'])"; done
python bench.py --deform-net --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deform-net', d['value'], d['ms_per_step'])"
python bench.py --autograd --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('autograd', d['value'], d['ms_per_step'])"
bash tools/profile_round.sh r01_t > /dev/null 2>&1; ls gpurun_out/r01_t
