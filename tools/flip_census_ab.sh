#!/bin/bash
# A/B flip census of the blend's exponent form: build libskgs_hip.so twice -- the product form (staged
# conic pre-scaled by log2 e, exponent straight into v_exp_f32) and the round-2 "e" form (-DSKGS_BLEND_LOG2E_PRESCALE=0:
# v_mul by log2 e + v_exp_f32) -- and run the full-size parity tests against the oracle with each.  The [census] lines
# say how many pixels took a different branch than the oracle, how far from the branch they were, and the worst error
# per tensor.  Run on the GPU box:  bash tools/flip_census_ab.sh > gpurun_out/flip_census_ab.txt 2>&1
set -u
cd "$(dirname "$0")/.."
mkdir -p _exp/obj
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
make -s -C sk_gs_amd/csrc
$HIPCC $FLAGS -DSKGS_BLEND_LOG2E_PRESCALE=0 -c sk_gs_amd/csrc/render.hip -o _exp/obj/render_e.o
OBJS=$(ls sk_gs_amd/csrc/_obj/*.o | grep -v '/render.o')
$HIPCC -shared -fPIC --offload-arch=gfx950 $OBJS _exp/obj/render_e.o -o _exp/libskgs_hip_e.so
for form in log2e e; do
  echo "==================== exponent form: $form"
  if [ $form = e ]; then export SKGS_HIP_LIB=$PWD/_exp/libskgs_hip_e.so; else unset SKGS_HIP_LIB; fi
  python -m pytest tests/test_gpu_fullsize.py -q -s -m gpu -k "full_size_parity_with_oracle" -p no:cacheprovider 2>&1 | grep -E "census|passed|failed|Error"
done
