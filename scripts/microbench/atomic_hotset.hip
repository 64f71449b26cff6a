// Microbenchmark: request rate of global float atomics as a function of the number of DISTINCT 64-B lines being hit
// (coarse hash-grid levels touch only (res+1)^3 entries: how much do hot lines cost?).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_hotset.hip -o atomic_hotset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// 4 adjacent lanes add to 4 consecutive floats of one line; the line is one of `hot` lines scattered over the buffer
__global__ void k_atomic(float* buf, uint32_t nslots, uint32_t hot, int reps, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t grp = tid >> 2, sub = tid & 3;
  for (int r = 0; r < reps; ++r) {
    uint32_t k = hash32(grp * 7919u + r * 104729u + seed) % hot;
    uint32_t slot = hash32(k * 2654435761u + 17u) % nslots;
    unsafeAtomicAdd(buf + (size_t)slot * 16 + sub, 1.0f);
  }
}
int main() {
  const size_t bytes = 64ull << 20;
  float* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
  uint32_t nslots = bytes / 64;
  const int threads = 1 << 22, reps = 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint32_t hot : {64u, 512u, 4913u, 12167u, 29791u, 79507u, 262144u, 1048576u}) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_atomic, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, hot, reps, 1234u + w);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_atomic, dim3(threads / 256), dim3(256), 0, 0, buf, nslots, hot, reps, 99u + w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double req = (double)threads * reps / 4;
    printf("hot lines %8u : %8.3f ms  %7.2f G requests/s\n", hot, ms, req / ms / 1e6);
  }
  return 0;
}
