// Microbenchmark: cost of an atomic wave-instruction as a function of how many of its 16 quads are active (run tails in k_grid_scatter
// leave 1-4 quads active per instruction).  Reports requests/s and wave-instructions/s.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_sparse.hip -o atomic_sparse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// quads q < active of every wave add (4 lanes of a quad -> 4 consecutive floats of one random line)
__global__ void k_atomic(float* buf, uint32_t nslots, int reps, uint32_t seed, int active) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t lane = threadIdx.x & 63, quad = lane >> 2, sub = lane & 3;
  // spread the active quads over the wave (quad index * 16 / active) like run tails are
  bool on = (quad * active) / 16 != ((quad + 1) * active) / 16 ? true : false;
  for (int r = 0; r < reps; ++r) {
    uint32_t slot = hash32((tid >> 2) * 7919u + r * 104729u + seed) % nslots;
    if (on) unsafeAtomicAdd(buf + (size_t)slot * 16 + sub, 1.0f);
  }
}
int main() {
  const size_t bytes = 64ull << 20;
  float* buf; (void)hipMalloc(&buf, bytes); (void)hipMemset(buf, 0, bytes);
  uint32_t nslots = bytes / 64;
  const int threads = 1 << 22, reps = 16;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int active : {16, 8, 4, 2, 1}) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_atomic, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, reps, 1234u + w, active);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_atomic, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, reps, 99u + w, active);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double instr = (double)threads / 64 * reps, req = instr * active;
    printf("active quads/instr %2d : %8.3f ms  %7.2f G requests/s  %6.2f G wave-instr/s  (%.0f cycles/instr/CU at 2.4 GHz)\n", active, ms, req / ms / 1e6,
           instr / ms / 1e6, ms * 1e-3 * 2.4e9 / (instr / 256));
  }
  return 0;
}
