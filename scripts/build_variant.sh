#!/bin/bash
# build_variant.sh <name> <file> <extra hipcc flags...>: nerfstudio-thermal_amd/build/libtn_<name>.so with <file>.hip compiled with the extra flags
# (timing diagnostics; select with TN_LIB=nerfstudio-thermal_amd/build/libtn_<name>.so)
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../nerfstudio-thermal_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result"
hipcc $FLAGS "$@" -c $file.hip -o ../build/${file}_$name.o
objs=""
for f in tn_misc tn_sampler tn_prop tn_field tn_scatter tn_splat tn_pipeline; do
  if [ $f == ${FILE_REPLACES:-$file} ]; then objs="$objs ../build/${file}_$name.o"; else objs="$objs ../build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../build/libtn_$name.so
echo built libtn_$name.so
