// global_load_lds_dwordx4 (gfx950): where the 64 lanes' 16 bytes land in LDS, and what a wave that streams an array through a private 1-KB LDS
// slot this way reaches -- the building block of "Adam inside the sampling chain's waves" (profiles/r06_next_sampling.md).
//   ./lds_dma            layout check (copy through LDS == input) + GB/s of the streamed copy for 1..4 arrays in flight per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ARR>
__global__ void __launch_bounds__(256) k_stream(const float* __restrict__ in, float* __restrict__ out, long n4_per_arr, int spin) {
  __shared__ __attribute__((aligned(16))) float stage[4 * ARR * 256];  // [wave][array][64 lanes x 4 floats]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* mine = stage + wv * ARR * 256;
  const long waves = (long)gridDim.x * 4, gw = (long)blockIdx.x * 4 + wv;
  float acc = 0.0f;
  for (long b = gw; b * 64 < n4_per_arr; b += waves) {
    const long g4 = b * 64 + lane;  // this lane's float4 of the batch
    if (g4 < n4_per_arr) {
#pragma unroll
      for (int a = 0; a < ARR; ++a)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + ((long)a * n4_per_arr + g4) * 4),
                                         (__attribute__((address_space(3))) void*)(mine + a * 256), 16, 0, 0);
    }
    for (int s = 0; s < spin; ++s) acc = acc * 1.0001f + 0.5f;  // stand-in for the work that hides the loads
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (g4 < n4_per_arr) {
#pragma unroll
      for (int a = 0; a < ARR; ++a) {
        v4f v = *reinterpret_cast<v4f*>(mine + a * 256 + lane * 4);
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(out + ((long)a * n4_per_arr + g4) * 4));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  if (acc == 1.2345e-30f) out[0] = acc;
}

template <int ARR>
static int run(const float* d_in, float* d_out, long n4, const std::vector<float>& h_in, int spin, bool check) {
  const long n4a = n4 / ARR;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMemset(d_out, 0, n4 * 16));
  hipLaunchKernelGGL(k_stream<ARR>, dim3(1024), dim3(256), 0, 0, d_in, d_out, n4a, spin);
  CK(hipDeviceSynchronize());
  if (check) {
    std::vector<float> h(n4a * ARR * 4);
    CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (size_t i = 0; i < h.size(); ++i) bad += h[i] != h_in[i];
    printf("arrays %d: layout %s (%ld of %zu floats differ)\n", ARR, bad ? "WRONG" : "ok", bad, h.size());
    if (bad) return 1;
  }
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_stream<ARR>, dim3(1024), dim3(256), 0, 0, d_in, d_out, n4a, spin);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("arrays %d spin %4d: %.1f us, read %.2f TB/s (+ the same written)\n", ARR, spin, ms * 1e3, n4a * ARR * 16.0 / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  const long n4 = 16l << 20;  // 16 M float4 = 256 MB
  std::vector<float> h(n4 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000003) * 0.25f;
  float *d_in, *d_out;
  CK(hipMalloc(&d_in, n4 * 16)); CK(hipMalloc(&d_out, n4 * 16));
  CK(hipMemcpy(d_in, h.data(), n4 * 16, hipMemcpyHostToDevice));
  if (run<1>(d_in, d_out, n4, h, 0, true)) return 1;
  if (run<4>(d_in, d_out, n4, h, 0, true)) return 1;
  for (int spin : {0, 200, 1000, 4000}) { if (run<4>(d_in, d_out, n4, h, spin, false)) return 1; }
  return 0;
}
