#!/bin/bash
# One gpurun call at a milestone: GPU tests (+ observed parity statistics), the default bench line, one bench line per BASELINE
# config, the dense variants, the forced 1-rank RCCL run, the comparison variants, kernel stats + PMC passes (config #1 and #4).
# usage (on the GPU box, from the repo root): bash tools/round_measurements.sh <tag>      e.g. r02_a
tag=${1:-rXX}
exec < /dev/null   # (nothing here reads stdin; a tool that does must not wait on the terminal of a batch run)
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
bash tools/gpu_tests.sh $tag ${SKGS_TEST_RUNS:-1}; echo "GPU TESTS rc=$?" | tee $out/gpu_tests_exit.txt   # whole log + rc kept
python bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-200 $out/bench_default.json
# stage sp (512 superpoints, 3+8-d search, sp_deform_net on 512 rows) at config #1's size, every weighting; with the CPU leg once
python bench.py --stage sp > $out/bench_stage_sp.json 2> $out/bench_stage_sp.err; cut -c1-200 $out/bench_stage_sp.json
for m in kernel dist W; do
  python bench.py --stage sp --lbs-method $m --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_stage_sp_$m.json 2>/dev/null
  python -c "import json; d=json.load(open('$out/bench_stage_sp_$m.json')); print('stage sp, $m', d['value'], d['ms_per_step'])"
done
python bench.py --stage sp --keep-order --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_stage_sp_keep-order.json 2>/dev/null
python bench.py --keep-order --steps 100 --warmup 10 --no-cpu-baseline --no-ms-per-render > $out/bench_keep-order.json 2>/dev/null
python -c "import json; a=json.load(open('$out/bench_stage_sp_keep-order.json')); b=json.load(open('$out/bench_keep-order.json')); print('unsorted Gaussians: stage sp', a['value'], ' stage sk', b['value'])"
python tools/time_sp_net.py 512 2>/dev/null > $out/time_sp_net.txt; cat $out/time_sp_net.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4x4_layer tools/micro/mfma4x4_layer.hip 2>/dev/null && /tmp/mfma4x4_layer 128 x | grep -v "    lane" > $out/mfma4x4_layer.txt 2>&1; tail -12 $out/mfma4x4_layer.txt
for c in 0 1 2 3 4; do
  python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_config$c.json 2>/dev/null
  python -c "import json; d=json.load(open('$out/bench_config$c.json')); print('config $c', d['value'], d['ms_per_step'])"
done
for sm in 2.5 4; do
  python bench.py --scale-mult $sm --steps 100 --warmup 10 --no-cpu-baseline --no-ms-per-render > $out/bench_dense_x$sm.json 2>/dev/null
  python -c "import json; d=json.load(open('$out/bench_dense_x$sm.json')); print('dense x$sm', d['value'], d['ms_per_step'], d['config']['num_rendered_mean'])"
done
SKGS_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_rccl_1rank.json
python -c "import json; d=json.load(open('$out/bench_rccl_1rank.json')); print('1-rank RCCL group', d['value'], d['ms_per_step'], d['config']['parallelism'])"
for v in "--bone-tables" "--layered-mlp" "--graph-per-view" "--autograd" "--compact-lists" "--serial-adam" "--fixed-joints"; do
  python bench.py $v --no-cpu-baseline --no-ms-per-render 2>/dev/null | tail -1 > "$out/bench_variant${v}.json"
  python -c "import json; d=json.load(open('$out/bench_variant${v}.json')); print('variant $v', d['value'], d['ms_per_step'])"
done
for c in 2 3 4; do
  python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-ms-per-render --pre-forward off 2>/dev/null | tail -1 > $out/bench_config${c}_pre-forward-off.json
  python -c "import json; d=json.load(open('$out/bench_config${c}_pre-forward-off.json')); print('config $c --pre-forward off', d['value'], d['ms_per_step'])"
done
SKGS_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-ms-per-render --pre-forward off 2>/dev/null | tail -1 > $out/bench_rccl_1rank_pre-forward-off.json
python -c "import json; d=json.load(open('$out/bench_rccl_1rank_pre-forward-off.json')); print('1-rank RCCL group --pre-forward off', d['value'], d['ms_per_step'])"
# per-rank cost of every exchange variant of the multi-rank step: the real RCCL backend with a 1-rank group (one GPU here)
port=29560
for x in allreduce factors factors-overlap pipeline allreduce-graph factors-graph factors-graph-split; do
  port=$((port+1))
  SKGS_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-ms-per-render --exchange $x 2>/dev/null | tail -1 > $out/bench_rccl_1rank_exchange_$x.json
  python -c "import json; d=json.load(open('$out/bench_rccl_1rank_exchange_$x.json')); print('1-rank RCCL group --exchange $x', d['value'], d['ms_per_step'])"
done
for c in 1 3; do
  python bench.py --config $c --steps 400 --warmup 50 --no-cpu-baseline --no-ms-per-render --densify-every 100 2>/dev/null | tail -1 > $out/bench_config${c}_densify-every-100.json
  python -c "import json; d=json.load(open('$out/bench_config${c}_densify-every-100.json')); print('config $c --densify-every 100', d['value'], d['ms_per_step'], d['densify']['ms'], d['densify']['P'], 'graphs', d['densify']['graphs_captured'])"
done
python tools/time_skeleton.py 2>/dev/null | grep -v "^$" > $out/time_skeleton.txt; tail -4 $out/time_skeleton.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_rate tools/micro/valu_issue_rate.hip 2>/dev/null && /tmp/valu_issue_rate > $out/valu_issue_rate.txt 2>&1; /tmp/valu_issue_rate >> $out/valu_issue_rate.txt 2>&1; tail -20 $out/valu_issue_rate.txt
python tools/time_mlp.py 2>/dev/null | grep "fused\|backward\|prologue" > $out/time_mlp.txt; cat $out/time_mlp.txt
python tools/time_densify.py 2>/dev/null | tail -2 > $out/time_densify.txt; cat $out/time_densify.txt
python tools/time_loss.py 2>/dev/null | tail -1 > $out/time_loss.txt; python tools/time_loss.py 1024 1024 2>/dev/null | tail -1 >> $out/time_loss.txt; cat $out/time_loss.txt
# operator path: compiled vs ctypes marshalling, backward on the calling thread vs torch's worker thread
for v in "1 caller" "1 worker" "0 caller" "0 worker"; do set -- $v
  SKGS_TORCH_OPS=$1 python bench.py --no-cpu-baseline --backward-thread $2 --steps 50 --warmup 10 2>/dev/null | tail -1 > $out/bench_oppath_ops$1_$2.json
  python -c "import json; d=json.load(open('$out/bench_oppath_ops$1_$2.json')); r=d['ms_per_render_fwd_bwd']; print('operator path: compiled marshalling $1, backward thread $2: ms/render', r['median'], 'replay', r['graph_replay_median'], 'kernels', r['kernel_sum'])"
done
# round 5: the schedule off, the fused step behind the autograd API, the shipped SC-GS / SP-GS combinations of stage sp, the unmodified
# reference's deform sequence through the lietorch / pytorch3d stand-ins, the one-round-latency sweep, pixels per lane
python bench.py --lr-schedule off --steps 200 --warmup 20 --no-cpu-baseline --no-ms-per-render --no-survey-recipe 2>/dev/null | tail -1 > $out/bench_lr-schedule-off.json
python bench.py --autograd-fused --steps 200 --warmup 20 --no-cpu-baseline --no-ms-per-render --no-survey-recipe 2>/dev/null | tail -1 > $out/bench_variant--autograd-fused.json
python -c "import json; a=json.load(open('$out/bench_lr-schedule-off.json')); b=json.load(open('$out/bench_variant--autograd-fused.json')); print('lr schedule off', a['value'], ' autograd-fused', b['value'])"
python bench.py --autograd-fused --serial-adam --steps 200 --warmup 20 --no-cpu-baseline --no-ms-per-render --no-survey-recipe 2>/dev/null | tail -1 > $out/bench_variant--autograd-fused_serial-adam.json
python bench.py --stage sp --raw-time --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_stage_sp_raw-time.json
python -c "import json; a=json.load(open('$out/bench_variant--autograd-fused_serial-adam.json')); b=json.load(open('$out/bench_stage_sp_raw-time.json')); print('autograd-fused + serial adam', a['value'], ' stage sp raw time', b['value'])"
for p in sc_gs sp_gs; do
  python bench.py --stage sp --preset $p --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_stage_sp_preset_$p.json
  python -c "import json; d=json.load(open('$out/bench_stage_sp_preset_$p.json')); print('stage sp preset $p', d['value'], d['ms_per_step'])"
done
for m in hooks accelerated fused; do
  timeout -k 5 300 python bench.py --reference-loop $m --steps 200 --warmup 10 2>/dev/null | tail -1 > $out/bench_reference-loop_$m.json
  python -c "import json; d=json.load(open('$out/bench_reference-loop_$m.json')); print('reference loop, $m', d['value'], d['ms_per_step'], d['config']['loss_last'])"
done
for m in hooks accelerated fused; do
  timeout -k 5 300 python bench.py --stage sp --reference-loop $m --steps 200 --warmup 10 2>/dev/null | tail -1 > $out/bench_stage_sp_reference-loop_$m.json
  python -c "import json; d=json.load(open('$out/bench_stage_sp_reference-loop_$m.json')); print('reference loop, stage sp, $m', d['value'], d['ms_per_step'], d['config']['loss_last'])"
done
# round 6: the fused route (sk_gs_amd/reference_fused.py) on the default line's scene, unprimed, with eager launches instead of graphs, stage sp
# with the W weighting (dense logit table under torch's Adam) and with the shipped weight regularisers (torch ops of the reference)
for v in "accelerated --loop-scene headline" "fused --loop-scene headline" "fused --prime-steps 0"; do set -- $v; m=$1; shift
  timeout -k 5 300 python bench.py --reference-loop $m "$@" --steps 200 --warmup 10 2>/dev/null | tail -1 > "$out/bench_reference-loop_${m}_${2:-x}${3:-}.json"
  python -c "import json; d=json.load(open('$out/bench_reference-loop_${m}_${2:-x}${3:-}.json')); print('reference loop, $v', d['value'], d['ms_per_step'], d['config']['loss_last'])"
done
SKGS_REF_FUSED_GRAPHS=0 timeout -k 5 300 python bench.py --reference-loop fused --steps 200 --warmup 10 2>/dev/null | tail -1 > $out/bench_reference-loop_fused_eager-launches.json
python -c "import json; d=json.load(open('$out/bench_reference-loop_fused_eager-launches.json')); print('reference loop, fused, SKGS_REF_FUSED_GRAPHS=0', d['value'], d['ms_per_step'])"
for v in "fused --lbs-method W" "accelerated --sp-regularisers" "fused --sp-regularisers" "fused --sp-regularisers --reg-torch" "fused --sp-regularisers --lbs-method W"; do set -- $v; m=$1; shift
  f="$out/bench_stage_sp_reference-loop_${m}$(echo "$@" | tr -d ' ' | sed 's/--/_/g').json"
  timeout -k 5 300 python bench.py --stage sp --reference-loop $m "$@" --steps 100 --warmup 10 2>/dev/null | tail -1 > "$f"
  python -c "import json; d=json.load(open('$f')); print('reference loop, stage sp, $v', d['value'], d['ms_per_step'], d['config']['loss_last'])"
done
SKGS_REF_TILED_ADAM=0 timeout -k 5 300 python bench.py --stage sp --reference-loop fused --lbs-method W --steps 100 --warmup 10 2>/dev/null | tail -1 > $out/bench_stage_sp_reference-loop_fused_lbs-methodW_dense-adam.json
python -c "import json; d=json.load(open('$out/bench_stage_sp_reference-loop_fused_lbs-methodW_dense-adam.json')); print('reference loop, stage sp, fused W, SKGS_REF_TILED_ADAM=0', d['value'], d['ms_per_step'])"
timeout -k 5 200 python tools/phase_times_reference_loop.py 2>/dev/null | tail -1 > $out/phase_times_reference_loop.txt; cat $out/phase_times_reference_loop.txt
timeout -k 5 200 python tools/find_host_spikes.py 2>/dev/null | tail -4 > $out/find_host_spikes.txt; cat $out/find_host_spikes.txt
timeout -k 5 300 python tools/preprocess_chain_sweep.py 2>/dev/null | grep -v amdgpu > $out/preprocess_chain_sweep.txt; cat $out/preprocess_chain_sweep.txt
timeout -k 5 200 python tools/time_reference_sequence.py 2>/dev/null | grep -v "not capturable" > $out/time_reference_sequence.txt; cat $out/time_reference_sequence.txt
timeout -k 5 300 python tools/round_latency_sweep.py 2>/dev/null | grep -v amdgpu > $out/round_latency_sweep.txt; cat $out/round_latency_sweep.txt
timeout -k 5 300 python tools/ppl_sweep.py 2>/dev/null | grep ppl > $out/ppl_sweep.txt; cat $out/ppl_sweep.txt
bash tools/profile_round.sh ${tag}_c1 > /dev/null 2>&1; ls gpurun_out/${tag}_c1 | head
bash tools/profile_round.sh ${tag}_c4 --config 4 > /dev/null 2>&1; ls gpurun_out/${tag}_c4 | head
bash tools/profile_round.sh ${tag}_sp --stage sp > /dev/null 2>&1; ls gpurun_out/${tag}_sp | head
