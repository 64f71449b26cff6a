// Microbenchmark: rate of LDS atomics per CU (what the fold pass of the binned scatter runs on): ds_add_f32 / ds_add_u32 / ds_add_rtn_u32 on
// random addresses of a 64-KB image vs conflict-free addresses vs a plain ds_read + ds_write pair.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate.hip -o atomic_lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE, int ADDR>
__global__ void __launch_bounds__(1024) k(float* out, int reps, int active) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int t = tid; t < 16384; t += blockDim.x) lds[t] = 0.0f;
  __syncthreads();
  uint32_t acc = 0;
  for (int r = 0; r < reps; ++r) {
    uint32_t a;
    if (ADDR == 0) a = hash32(tid * 7919u + r * 104729u + blockIdx.x) & 16383u;   // random word
    else if (ADDR == 1) a = ((tid + r * 64) & 16383u);                              // conflict-free, consecutive
    else a = (hash32((tid >> 6) * 31u + r) & 255u) * 64u + (tid & 63);             // conflict-free, random row
    if (MODE == 0) { if ((int)(hash32(tid + r * 977u) & 63u) < active) unsafeAtomicAdd(&lds[a], 1.0f); }
    else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(lds) + a, 1u);
    else if (MODE == 2) acc += atomicAdd(reinterpret_cast<uint32_t*>(lds) + a, 1u);
    else if (MODE == 3) { float v = lds[a]; lds[a] = v + 1.0f; }
    else if (MODE == 4) unsafeAtomicAdd(reinterpret_cast<double*>(lds) + (a >> 1), 1.0);                          // ds_add_f64 on 8192 doubles
    else if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + (a >> 1), 1ull);                   // ds_add_u64
    else if (MODE == 6) { unsafeAtomicAdd(&lds[a & ~1u], 1.0f); unsafeAtomicAdd(&lds[a | 1u], 1.0f); }            // the two components of a float2 slot
    else if (MODE == 7) { unsafeAtomicAdd(reinterpret_cast<double*>(lds) + ((a >> 2) << 1), 1.0); unsafeAtomicAdd(reinterpret_cast<double*>(lds) + ((a >> 2) << 1) + 1, 1.0); }
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = lds[5] + (float)acc;
}
template <int MODE, int ADDR>
void run(const char* name, float* out, int threads, int active = 64) {
  const int blocks = 512, reps = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, ADDR>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<MODE, ADDR>), dim3(blocks), dim3(threads), 65536, 0, out, reps, active);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<MODE, ADDR>), dim3(blocks), dim3(threads), 65536, 0, out, reps, active);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double lane_ops = (double)blocks * threads * reps;
  // 512 blocks of 64 KB LDS: 2 per CU resident -> every CU busy
  printf("%-34s act %2d threads %4d: %8.3f ms  %7.1f G lane-ops/s chip  %6.2f cycles per wave-instr per CU\n", name, active, threads, ms, lane_ops / ms / 1e6,
         ms * 1e-3 * 2.4e9 / (lane_ops / 64 / 256));
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  for (int threads : {256, 1024}) {
    run<0, 0>("ds_add_f32 random", out, threads);
    run<0, 1>("ds_add_f32 consecutive", out, threads);
    for (int act : {32, 16, 8, 4, 1}) run<0, 0>("ds_add_f32 random, sparse lanes", out, threads, act);
    run<0, 2>("ds_add_f32 conflict-free rows", out, threads);
    run<1, 0>("ds_add_u32 random", out, threads);
    run<1, 1>("ds_add_u32 consecutive", out, threads);
    run<2, 0>("ds_add_rtn_u32 random", out, threads);
    run<2, 1>("ds_add_rtn_u32 consecutive", out, threads);
    run<4, 0>("ds_add_f64 random", out, threads);
    run<4, 1>("ds_add_f64 consecutive", out, threads);
    run<5, 0>("ds_add_u64 random", out, threads);
    run<6, 0>("2 x ds_add_f32 (float2 slot) random", out, threads);
    run<7, 0>("2 x ds_add_f64 (double2 slot) random", out, threads);
    run<3, 0>("ds_read+ds_write random", out, threads);
    run<3, 1>("ds_read+ds_write consecutive", out, threads);
  }
  return 0;
}
