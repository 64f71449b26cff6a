#!/bin/bash
# builds nerfstudio-thermal_amd/build/libtn_ablate<N>.so with -DFB_ABLATE=N (timing diagnostics of k_field_bwd_fused; TN_LIB selects it)
set -e
cd "$(dirname "$0")/../nerfstudio-thermal_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result"
for n in "$@"; do
  hipcc $FLAGS -DFB_ABLATE=$n -c tn_field.hip -o ../build/tn_field_abl$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC ../build/tn_misc.o ../build/tn_sampler.o ../build/tn_prop.o ../build/tn_field_abl$n.o ../build/tn_scatter.o ../build/tn_splat.o ../build/tn_pipeline.o -o ../build/libtn_ablate$n.so
done
