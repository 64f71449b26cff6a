// Microbenchmark: rate of random 8-byte gathers (one hash-table corner each) as a function of the table size -- what bounds the hash-grid
// forward.  Every thread issues GROUP independent fetches before it uses any of them (as k_field_fwd_fused does with its 64 corners), the
// addresses come from an integer hash of (thread, round), so every fetch touches its own 64-B line with high probability.
// build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int GROUP>
__global__ void __launch_bounds__(256) k(const float2* __restrict__ t, uint32_t mask, int rounds, float* out) {
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    float2 v[GROUP];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) v[g] = t[hash32(tid * 0x9E3779B9u + (uint32_t)(r * GROUP + g) * 0x85EBCA6Bu) & mask];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) acc += v[g].x + v[g].y;
  }
  if (acc == 123.456f) out[0] = acc;
}
template <int GROUP>
static void run(const float2* t, size_t entries, float* out, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, rounds = 512 / GROUP * 4;  // 256 CUs x (4 SIMDs x waves_per_simd waves) = blocks of 4 waves
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<GROUP>, dim3(blocks), dim3(256), 0, 0, t, (uint32_t)(entries - 1), rounds, out);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<GROUP>, dim3(blocks), dim3(256), 0, 0, t, (uint32_t)(entries - 1), rounds, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double n = (double)blocks * 256 * rounds * GROUP;
  printf("table %6.0f MB  %2d fetches in flight per lane  %d waves/SIMD: %7.3f ms  %6.1f G gathers/s  = %5.2f TB/s of 64-B lines\n", entries * 8 / 1048576.0, GROUP,
         waves_per_simd, ms, n / ms / 1e6, n * 64 / ms / 1e9);
}
int main() {
  float* out; (void)hipMalloc(&out, 64);
  for (size_t mb : {4, 64, 128, 512, 2048}) {
    const size_t entries = mb * 1048576 / 8;
    float2* t; (void)hipMalloc(&t, entries * 8); (void)hipMemset(t, 0, entries * 8);
    run<8>(t, entries, out, 8);
    run<8>(t, entries, out, 2);
    run<64>(t, entries, out, 2);
    (void)hipFree(t);
  }
  return 0;
}
