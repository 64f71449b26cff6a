// Microbenchmark: does vector-ALU / LDS work overlap with v_mfma_f32_32x32x2_f32 on one SIMD of gfx950?
//   (a) same wave: a dependent MFMA chain with K INDEPENDENT v_fma_f32 written between consecutive MFMAs (K = 0, 4, 8, 12, 15, 24);
//   (b) two waves per SIMD: one runs the MFMA chain, the other only v_fma chains (the same number of instructions the (a) case interleaves);
//   (c) same wave, K ds_read_b32 between consecutive MFMAs.
// If the matrix pipe ran beside the vector ALU, the time of (a) would stay at the MFMA-only time up to K ~ 15 (15 x 4 cycles < 64 cycles per MFMA),
// and (b) would cost max(MFMA, VALU).  The fused MLP kernels of this package interleave ~4-7 vector / LDS instructions per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__global__ void __launch_bounds__(256) k_same_wave(float* out, int iters, float a, float b) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = (float)threadIdx.x;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);  // 8 independent chains
      __builtin_amdgcn_sched_barrier(0);  // keep the written order: MFMA, K x v_fma, MFMA, ...
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c[r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[0] = s;
}

// role by wave parity inside the block: waves 0..3 (one per SIMD) run MFMAs, waves 4..7 (the second wave of each SIMD) run K v_fma per "slot"
template <int K>
__global__ void __launch_bounds__(512) k_two_waves(float* out, int iters, float a, float b) {
  const bool mf = (threadIdx.x >> 6) < 4;
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = (float)threadIdx.x;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  if (mf) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  } else {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int k = 0; k < K; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c[r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[0] = s;
}

template <int K>
__global__ void __launch_bounds__(256) k_same_wave_lds(float* out, int iters, float a, float b) {
  __shared__ float sh[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = (float)i;
  __syncthreads();
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = (float)threadIdx.x;
  float acc = 0.f;
  int idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) acc += sh[(idx + 64 * k + 17 * u) & 4095];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c[r];
  if (s == 123.456f) out[0] = s;
}

template <typename Kn>
static float run(Kn kern, int threads, float* out, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
  const int iters = 2000;
  const double mfmas = iters * 16.0;  // per wave
  auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / mfmas; };
  printf("one wave per SIMD, MFMA chain with K independent v_fma_f32 between consecutive MFMAs (cycles per MFMA slot at 2.4 GHz):\n");
#define A(K) { float ms = run(k_same_wave<K>, 256, out, iters); printf("  K = %2d: %7.3f ms = %6.1f cycles per slot\n", K, ms, cyc(ms)); }
  A(0) A(4) A(8) A(12) A(15) A(24)
  printf("two waves per SIMD: one runs the MFMA chain, the other K v_fma_f32 per slot:\n");
#define B(K) { float ms = run(k_two_waves<K>, 512, out, iters); printf("  K = %2d: %7.3f ms = %6.1f cycles per slot\n", K, ms, cyc(ms)); }
  B(0) B(4) B(8) B(12) B(15) B(24)
  printf("one wave per SIMD, MFMA chain with K ds_read_b32 (+ v_add) between consecutive MFMAs:\n");
#define C(K) { float ms = run(k_same_wave_lds<K>, 256, out, iters); printf("  K = %2d: %7.3f ms = %6.1f cycles per slot\n", K, ms, cyc(ms)); }
  C(0) C(2) C(4) C(8)
  return 0;
}
