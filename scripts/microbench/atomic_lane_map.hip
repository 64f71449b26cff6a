// Microbenchmark: does the 64-B request coalescing of global float atomics need the 4 lanes of a line to be ADJACENT lanes, or do lanes
// 16 apart (same position in the four 16-lane rows) coalesce as well?  (Decides whether the scatter's segmented scans can stay inside
// a DPP row.)  Result: identical with all 64 lanes active -- but in k_grid_scatter only run-tail samples are active, and there the row
// layout lost 20 % (four active quads per sample instead of one): the real kernel keeps the adjacent-lane layout.   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_lane_map.hip -o atomic_lane_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// MODE 0: sample = lane>>2, sub = lane&3 (adjacent lanes share a line); MODE 1: sample = lane&15, sub = lane>>4 (lanes 16 apart share a line)
template <int MODE>
__global__ void k_atomic(float* buf, uint32_t nslots, int reps, uint32_t seed) {
  uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t smp = MODE == 0 ? (lane >> 2) : (lane & 15), sub = MODE == 0 ? (lane & 3) : (lane >> 4);
  for (int r = 0; r < reps; ++r) {
    uint32_t slot = hash32((wave * 16 + smp) * 7919u + r * 104729u + seed) % nslots;
    unsafeAtomicAdd(buf + (size_t)slot * 16 + sub, 1.0f);
  }
}
int main() {
  const size_t bytes = 64ull << 20;
  float* buf; (void)hipMalloc(&buf, bytes); (void)hipMemset(buf, 0, bytes);
  uint32_t nslots = bytes / 64;
  const int threads = 1 << 22, reps = 16;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, reps, 1234u + w);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, reps, 99u + w);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %8.3f ms  %7.2f G lines/s\n", name, ms, (double)threads * reps / 4 / ms / 1e6);
  };
  run(k_atomic<0>, "4 adjacent lanes per line");
  run(k_atomic<1>, "4 lanes 16 apart per line (one per DPP row)");
  return 0;
}
