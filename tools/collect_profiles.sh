#!/bin/bash
# Copy the summaries of one tools/round_measurements.sh run (gpurun_out/<tag>, gpurun_out/<tag>_c1, gpurun_out/<tag>_c4) into
# profiles/ under the round's naming; the PMC tables are rebuilt (config #4 first: pmc_render_backward.json is config #1's).
# usage (here, after the gpurun call has merged its output): bash tools/collect_profiles.sh r03_i
tag=$1
src=gpurun_out/$tag
[ -d "$src" ] || { echo "no $src"; exit 1; }
for f in $src/bench_*.json; do
  b=$(basename "$f" .json); b=${b//--/_}; b=${b// /_}
  [ -s "$f" ] && cp "$f" "profiles/${tag}_${b}.json"
done
for f in gpu_tests.txt gpu_tests_rc.txt gpu_tests_exit.txt parity_observed.json time_densify.txt time_mlp.txt time_skeleton.txt time_loss.txt valu_issue_rate.txt time_sp_net.txt mfma4x4_layer.txt time_reference_sequence.txt round_latency_sweep.txt ppl_sweep.txt phase_times_reference_loop.txt find_host_spikes.txt preprocess_chain_sweep.txt; do
  [ -s "$src/$f" ] && cp "$src/$f" "profiles/${tag}_$f"
done
d=gpurun_out/${tag}_sp   # stage sp: kernel stats + the counters' table (its own file: pmc_render_backward.json stays config #1 / stage sk)
if [ -d "$d" ]; then
  cp $d/bench_steps100.json profiles/${tag}_sp_bench_steps100.json
  cp $(ls -t $d/stats/*/*kernel_stats.csv | head -1) profiles/${tag}_sp_kernel_stats_bench_steps100.csv
  SKGS_PMC_JSON=pmc_render_backward_sp.json SKGS_PROFILE_CONFIG="hook-like-100k-800, stage sp" python tools/pmc_summary.py ${tag}_sp $d/fetch $d/write $d/valu > /dev/null
fi
for c in c4 c1; do
  d=gpurun_out/${tag}_$c
  [ -d "$d" ] || continue
  cp $d/bench_steps100.json profiles/${tag}_${c}_bench_steps100.json
  cp $(ls -t $d/stats/*/*kernel_stats.csv | head -1) profiles/${tag}_${c}_kernel_stats_bench_steps100.csv
  if [ $c = c4 ]; then title="config #4: 500k Gaussians, 24 joints, 1024x1024"; cfg=zju-like-500k-1024; else title="config #1: 100k Gaussians, 20 bones, 800x800"; cfg=hook-like-100k-800; fi
  SKGS_PROFILE_TITLE="$title" SKGS_PROFILE_CONFIG="$cfg" python tools/pmc_summary.py ${tag}_$c $d/fetch $d/write $d/valu > /dev/null
  [ $c = c4 ] && cp profiles/pmc_render_backward.json profiles/${tag}_c4_pmc_render_backward.json
done
ls profiles | grep "^$tag" | wc -l
