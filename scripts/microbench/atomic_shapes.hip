// Microbenchmark: request-rate of global float atomics by lane grouping (how many adjacent lanes share one 64-B line).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_shapes.hip -o atomic_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// GROUP adjacent lanes add to GROUP consecutive floats (one random 64-B-aligned..16B aligned slot per group); REPS instructions per lane
template <int GROUP>
__global__ void k_atomic(float* buf, uint32_t nslots, int reps, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t grp = tid / GROUP, sub = tid % GROUP;
  for (int r = 0; r < reps; ++r) {
    uint32_t slot = hash32(grp * 7919u + r * 104729u + seed) % nslots;   // slot = 16 floats (64 B)
    float* p = buf + (size_t)slot * 16 + sub;
    unsafeAtomicAdd(p, 1.0f);
  }
}
int main() {
  const size_t bytes = 64ull << 20;  // 64 MiB table like the main hash grid gradient
  float* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
  uint32_t nslots = bytes / 64;
  const int threads = 1 << 22, reps = 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int group) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, reps, 1234u + w);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, reps, 99u + w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double lanes = (double)threads * reps;
    printf("%-28s %8.3f ms  %7.2f G lane-atomics/s  %7.2f G requests/s (lanes/group)\n", name, ms, lanes / ms / 1e6, lanes / group / ms / 1e6);
  };
  run(k_atomic<1>, "1 lane / 64B line", 1);
  run(k_atomic<2>, "2 adjacent lanes / line", 2);
  run(k_atomic<4>, "4 adjacent lanes / line", 4);
  run(k_atomic<8>, "8 adjacent lanes / line", 8);
  run(k_atomic<16>, "16 adjacent lanes / line", 16);
  return 0;
}
