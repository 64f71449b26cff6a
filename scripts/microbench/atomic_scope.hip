// Microbenchmark: do narrower-scope float atomics execute in the XCD-local L2 (no EA request) when each XCD owns its replica?
// Workgroups are dispatched round-robin over the 8 XCDs, so replica = blockIdx.x % 8 is private to one XCD's L2.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_scope.hip -o atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int SCOPE, bool PRIVATE>
__global__ void k_atomic(float* buf, uint32_t lines_per_replica, int reps, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t grp = tid >> 2, sub = tid & 3;
  for (int r = 0; r < reps; ++r) {
    uint32_t h = hash32(grp * 7919u + r * 104729u + seed);
    uint32_t rep = PRIVATE ? (blockIdx.x & 7u) : (h >> 20) & 7u;
    uint32_t line = h % lines_per_replica;
    float* p = buf + ((size_t)rep * lines_per_replica + line) * 16 + sub;
    __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, SCOPE);
  }
}
int main() {
  const int threads = 1 << 22, reps = 8;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (uint32_t lines : {2048u, 16384u, 65536u, 262144u}) {   // per replica: 128 KB, 1 MB, 4 MB, 16 MB
    size_t bytes = (size_t)8 * lines * 64;
    float* buf; (void)hipMalloc(&buf, bytes);
    auto run = [&](auto kern, const char* name) {
      (void)hipMemset(buf, 0, bytes);
      hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, buf, lines, reps, 1234u);
      (void)hipEventRecord(e0);
      for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(kern, dim3(threads / 256), dim3(256), 0, 0, buf, lines, reps, 99u + w);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
      std::vector<float> h(bytes / 4);
      (void)hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost);
      double sum = 0; for (float v : h) sum += v;
      double expect = 5.0 * threads * reps;
      printf("lines/replica %7u  %-34s %7.3f ms  %6.2f G requests/s  sum %s (%.0f vs %.0f)\n", lines, name, ms, (double)threads * reps / 4 / ms / 1e6,
             sum == expect ? "OK" : "MISMATCH", sum, expect);
    };
    run(k_atomic<__HIP_MEMORY_SCOPE_AGENT, true>, "agent scope, XCD-private replica");
    run(k_atomic<__HIP_MEMORY_SCOPE_WORKGROUP, true>, "workgroup scope, XCD-private replica");
    run(k_atomic<__HIP_MEMORY_SCOPE_WAVEFRONT, true>, "wavefront scope, XCD-private");
    run(k_atomic<__HIP_MEMORY_SCOPE_AGENT, false>, "agent scope, shared replicas");
    run(k_atomic<__HIP_MEMORY_SCOPE_WORKGROUP, false>, "workgroup scope, shared (racy?)");
    (void)hipFree(buf);
  }
  return 0;
}
