// Microbenchmark: sustained rate of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 (the fp32 matrix instructions of the field kernels) with
// 1, 2 and 4 waves per SIMD and 1, 2 or 4 independent accumulators per wave: the floor every "MFMA-bound" claim in DESIGN.md is priced against.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a, float b) {
  f32x16 c[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[i][r] = (float)threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c[i][r];
  if (s == 123.456f) out[0] = s;
}
template <int ACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a, float b) {
  f32x4 c[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[i][r] = (float)threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += c[i][r];
  if (s == 123.456f) out[0] = s;
}
template <typename K>
static void run(const char* name, K kern, int acc, int waves_per_simd, double flops_per_mfma, float* out) {
  const int iters = 2000, blocks = 256 * waves_per_simd;  // blocks of 4 waves: one wave per SIMD each
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 4 * iters * 16 * acc;  // MFMAs issued
  printf("%-22s %d accumulators, %d waves/SIMD: %7.3f ms  %6.1f TFLOP/s  %6.1f ns per MFMA and SIMD (= %5.1f cycles at 2.4 GHz)\n", name, acc, waves_per_simd, ms,
         n * flops_per_mfma / ms / 1e9, ms * 1e6 / (n / 1024.0), ms * 1e6 / (n / 1024.0) * 2.4);
}
int main() {
  float* out; (void)hipMalloc(&out, 64);
  for (int w : {1, 2, 4}) {
    run("mfma_f32_32x32x2_f32", k32<1>, 1, w, 4096.0, out);
    run("mfma_f32_32x32x2_f32", k32<2>, 2, w, 4096.0, out);
    run("mfma_f32_32x32x2_f32", k32<4>, 4, w, 4096.0, out);
  }
  for (int w : {1, 2}) {
    run("mfma_f32_16x16x4_f32", k16<1>, 1, w, 2048.0, out);
    run("mfma_f32_16x16x4_f32", k16<4>, 4, w, 2048.0, out);
  }
  return 0;
}
