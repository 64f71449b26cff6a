// Calibration kernels for the HBM-side counters on gfx950 (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves a KNOWN number of bytes in one of the access shapes of libthermal_nerf_hip; scripts/pmc_summary.py divides the known bytes
// by what FETCH_SIZE / TCC_EA0_RDREQ / WRITE_SIZE reported for the kernel and applies that factor to the library's kernels of the same shape.
//   calib_read16   : streaming read, 16 B per lane (float4)      -- Adam, fold records, activation tiles
//   calib_read8    : streaming read,  8 B per lane (float2)      -- g_enc rows per level in the bin pass
//   calib_gather8  : random 8-B gather from a table >> Infinity Cache (512 MB), one 64-B line per lane   -- hash-table corners (upper bound per fetch)
//   calib_write16  : streaming write, 16 B per lane
//   calib_write10  : 8-B + 2-B stores to two arrays (the scatter's (value, slot) records)
// Buffers are 1 GiB so that nothing is served from the 256 MiB Infinity Cache between launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void calib_read16(const float4* __restrict__ p, int64_t n, float* out) {
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) *out = acc;
}
__global__ void calib_read8(const float2* __restrict__ p, int64_t n, float* out) {
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { float2 v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678f) *out = acc;
}
__global__ void calib_gather8(const float2* __restrict__ p, int64_t entries, int64_t n, float* out) {
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    float2 v = p[h % (uint64_t)entries];
    acc += v.x + v.y;
  }
  if (acc == 12345.678f) *out = acc;
}
__global__ void calib_write16(float4* __restrict__ p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void calib_write10(float2* __restrict__ v, uint16_t* __restrict__ s, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { v[i] = make_float2(1.f, (float)i); s[i] = (uint16_t)i; }
}
int main() {
  const int64_t bytes = 1ll << 30;
  void *a = nullptr, *b = nullptr; float* out = nullptr;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  const int grid = 256 * 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_read16, dim3(grid), dim3(256), 0, 0, (const float4*)a, bytes / 16, out);
    hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (float4*)b, bytes / 16);
    hipLaunchKernelGGL(calib_read8, dim3(grid), dim3(256), 0, 0, (const float2*)a, bytes / 8, out);
    hipLaunchKernelGGL(calib_write10, dim3(grid), dim3(256), 0, 0, (float2*)b, (uint16_t*)a, (int64_t)(64 << 20));
    hipLaunchKernelGGL(calib_gather8, dim3(grid), dim3(256), 0, 0, (const float2*)b, (int64_t)(bytes / 2 / 8), (int64_t)(32 << 20), out);
    CK(hipDeviceSynchronize());
  }
  // known bytes per launch (printed for the record; pmc_summary.py carries the same numbers)
  printf("calib_read16 %lld\ncalib_write16 %lld\ncalib_read8 %lld\ncalib_write10 %lld\ncalib_gather8_lines64 %lld\n", (long long)bytes, (long long)bytes,
         (long long)bytes, (long long)(64ll << 20) * 10, (long long)(32ll << 20) * 64);
  return 0;
}
