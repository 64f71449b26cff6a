#!/bin/bash
# Builds libthermal_nerf_hip.so for gfx950 (no GPU needed: hipcc cross-compiles).
set -e
cd "$(dirname "$0")"
OUT=../libthermal_nerf_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result"
mkdir -p ../build
pids=()
for f in tn_misc tn_sampler tn_prop tn_field tn_scatter tn_splat tn_pipeline tn_comm; do
  hipcc $FLAGS -c $f.hip -o ../build/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map ../build/tn_misc.o ../build/tn_sampler.o ../build/tn_prop.o ../build/tn_field.o ../build/tn_scatter.o ../build/tn_splat.o ../build/tn_pipeline.o ../build/tn_comm.o -ldl -o $OUT
echo "built $OUT"
