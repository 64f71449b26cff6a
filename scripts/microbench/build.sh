#!/bin/bash
# Builds the microbenchmarks (atomic shapes / scopes, LDS atomic rate, PMC calibration, XCD-affine encode mappings, fp32 MFMA
# issue rate, hash-table gather rate) for gfx950 next to their sources (cross-compiles without a GPU); run them on the GPU box, e.g.
#   gpurun -- 'timeout 120 scripts/microbench/atomic_shapes'
set -e
cd "$(dirname "$0")"
for f in atomic_*.hip lds_*.hip pmc_*.hip encode_*.hip mfma_*.hip gather_*.hip; do
  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -Wno-unused-result "$f" -o "${f%.hip}"
done
ls -1 atomic_* lds_* pmc_* encode_* mfma_* gather_* | grep -v '\.hip$'
