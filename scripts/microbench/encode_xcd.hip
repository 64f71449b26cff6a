// Microbenchmark: 16-level hash-grid encode of 4096 x 48 samples, two mappings.
//   A  "fused mapping": lane = (sample, half of the levels), 64 corner fetches per lane in flight -- what k_field_fwd_fused does.  Every CU
//      gathers from all 16 levels: the 11 hashed levels (46 MB) do not fit any L2 (4 MB per XCD).
//   B  "XCD-affine, level-major": block b runs on XCD b % 8 (round-robin dispatch) and gathers from ONE level at a time, chosen so that an XCD
//      only ever touches levels {x, x + 8}: the level's 4 MB table stays in that XCD's L2.  thread = sample; output level-major [L][P] float2.
// Positions: rays in 2x2-pixel patches towards the scene centre, 48 sorted samples per ray in the reference's lin-disparity spacing,
// L-inf scene contraction -> [0,1]^3 (same arithmetic as tn_common.h).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off encode_xcd.hip -o encode_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u
struct Corners { uint32_t idx[8]; float ox, oy, oz; };
__device__ __forceinline__ void corners(float px, float py, float pz, float res, uint32_t mask, uint32_t off, Corners& lc) {
  float sx = px * res, sy = py * res, sz = pz * res;
  float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
  uint32_t cx = (uint32_t)(int)ceilf(sx), cy = (uint32_t)(int)ceilf(sy), cz = (uint32_t)(int)ceilf(sz);
  uint32_t fx = (uint32_t)(int)fxf, fy = (uint32_t)(int)fyf, fz = (uint32_t)(int)fzf;
  lc.ox = sx - fxf; lc.oy = sy - fyf; lc.oz = sz - fzf;
  uint32_t hcy = cy * PRIME_Y, hfy = fy * PRIME_Y, hcz = cz * PRIME_Z, hfz = fz * PRIME_Z;
  lc.idx[0] = ((cx ^ hcy ^ hcz) & mask) + off; lc.idx[1] = ((cx ^ hfy ^ hcz) & mask) + off;
  lc.idx[2] = ((fx ^ hfy ^ hcz) & mask) + off; lc.idx[3] = ((fx ^ hcy ^ hcz) & mask) + off;
  lc.idx[4] = ((cx ^ hcy ^ hfz) & mask) + off; lc.idx[5] = ((cx ^ hfy ^ hfz) & mask) + off;
  lc.idx[6] = ((fx ^ hfy ^ hfz) & mask) + off; lc.idx[7] = ((fx ^ hcy ^ hfz) & mask) + off;
}
__device__ __forceinline__ float2 interp(const float2 f[8], float ox, float oy, float oz) {
  float ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
  float2 r;
#define L3(C) { float a = f[0].C * ox + f[3].C * ux, b = f[1].C * ox + f[2].C * ux, c = f[5].C * ox + f[6].C * ux, d = f[4].C * ox + f[7].C * ux; \
                float e = a * oy + b * uy, g = d * oy + c * uy; r.C = e * oz + g * uz; }
  L3(x) L3(y)
#undef L3
  return r;
}
struct Res { float r[16]; };

__global__ void __launch_bounds__(256) k_fused_map(const float2* __restrict__ table, const float* __restrict__ pos, int64_t P, Res R, uint32_t mask,
                                                   uint32_t tsize, float4* __restrict__ enc) {
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int64_t ntiles = (P + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    int64_t p = tile * 32 + j; if (p >= P) p = P - 1;
    const float px = pos[3 * p], py = pos[3 * p + 1], pz = pos[3 * p + 2];
    Corners lc[8]; float2 fv[8][8];
#pragma unroll
    for (int qi = 0; qi < 8; ++qi) { const int l = 4 * (qi >> 1) + 2 * h + (qi & 1); corners(px, py, pz, R.r[l], mask, (uint32_t)l * tsize, lc[qi]); }
#pragma unroll
    for (int qi = 0; qi < 8; ++qi)
#pragma unroll
      for (int k = 0; k < 8; ++k) fv[qi][k] = table[lc[qi].idx[k]];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float2 a = interp(fv[2 * q], lc[2 * q].ox, lc[2 * q].oy, lc[2 * q].oz), b = interp(fv[2 * q + 1], lc[2 * q + 1].ox, lc[2 * q + 1].oy, lc[2 * q + 1].oz);
      enc[(tile * 4 + q) * 64 + lane] = make_float4(a.x, a.y, b.x, b.y);
    }
  }
}

// XCD-affine: blockIdx -> (xcd = b % 8, q = b / 8); the q range is split into `phases` consecutive parts, part ph handles level xcd + 8 * ph
// (L = 16: two phases).  PER = samples per thread.
template <int PER>
__global__ void __launch_bounds__(256) k_xcd_map(const float2* __restrict__ table, const float* __restrict__ pos, int64_t P, Res R, uint32_t mask,
                                                 uint32_t tsize, int chunks, float2* __restrict__ enc) {
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ph = q / chunks, chunk = q - ph * chunks;
  const int l = xcd + 8 * ph;
  const float res = R.r[l];
  const uint32_t off = (uint32_t)l * tsize;
  const int64_t base = (int64_t)chunk * 256 * PER + threadIdx.x;
  Corners lc[PER]; float2 fv[PER][8]; int64_t pp[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    int64_t p = base + u * 256; pp[u] = p; if (p >= P) p = P - 1;
    corners(pos[3 * p], pos[3 * p + 1], pos[3 * p + 2], res, mask, off, lc[u]);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u)
#pragma unroll
    for (int k = 0; k < 8; ++k) fv[u][k] = table[lc[u].idx[k]];
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (pp[u] < P) enc[(int64_t)l * P + pp[u]] = interp(fv[u], lc[u].ox, lc[u].oy, lc[u].oz);
}


// C: B with a LANE PAIR per sample: the even lane fetches the four corners with x = ceil, the odd lane those with x = floor -- the two x
// neighbours of a corner pair sit in the same 64-B line 7 times out of 8, and now they are adjacent lanes of ONE load instruction.  The pair
// swaps one component per corner over DPP and each lane interpolates one of the two features.
template <int MODE>
__device__ __forceinline__ float2 ld8(const float2* p) {
  if (MODE == 1) { typedef float v2 __attribute__((ext_vector_type(2))); v2 t = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p)); return make_float2(t.x, t.y); }
  if (MODE == 2) { unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                   return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }
  if (MODE == 3) { unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                   return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))); }
  return *p;
}
template <int PER, int MODE = 0>
__global__ void __launch_bounds__(256) k_xcd_pair(const float2* __restrict__ table, const float* __restrict__ pos, int64_t P, Res R, uint32_t mask,
                                                  uint32_t tsize, int chunks, float* __restrict__ enc) {
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ph = q / chunks, chunk = q - ph * chunks;
  const int l = xcd + 8 * ph;
  const float res = R.r[l];
  const uint32_t off = (uint32_t)l * tsize;
  const int side = threadIdx.x & 1;
  const int64_t base = (int64_t)chunk * 128 * PER + (threadIdx.x >> 1);
  float2 fv[PER][4]; float ox[PER], oy[PER], oz[PER]; int64_t pp[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    int64_t p = base + u * 128; pp[u] = p; if (p >= P) p = P - 1;
    const float sx = pos[3 * p] * res, sy = pos[3 * p + 1] * res, sz = pos[3 * p + 2] * res;
    const float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
    const uint32_t x = side ? (uint32_t)(int)fxf : (uint32_t)(int)ceilf(sx);
    const uint32_t hcy = (uint32_t)(int)ceilf(sy) * PRIME_Y, hfy = (uint32_t)(int)fyf * PRIME_Y, hcz = (uint32_t)(int)ceilf(sz) * PRIME_Z, hfz = (uint32_t)(int)fzf * PRIME_Z;
    ox[u] = sx - fxf; oy[u] = sy - fyf; oz[u] = sz - fzf;
    fv[u][0] = ld8<MODE>(table + ((x ^ hcy ^ hcz) & mask) + off); fv[u][1] = ld8<MODE>(table + ((x ^ hfy ^ hcz) & mask) + off);
    fv[u][2] = ld8<MODE>(table + ((x ^ hcy ^ hfz) & mask) + off); fv[u][3] = ld8<MODE>(table + ((x ^ hfy ^ hfz) & mask) + off);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const float wk = side ? 1.0f - ox[u] : ox[u], wr = side ? ox[u] : 1.0f - ox[u];
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float keep = side ? fv[u][k].y : fv[u][k].x, send = side ? fv[u][k].x : fv[u][k].y;
      const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
      t[k] = keep * wk + recv * wr;
    }
    const float uy = 1.0f - oy[u], uz = 1.0f - oz[u];
    const float e = t[0] * oy[u] + t[1] * uy, g = t[2] * oy[u] + t[3] * uy;
    if (pp[u] < P) enc[((int64_t)l * P + pp[u]) * 2 + side] = e * oz[u] + g * uz;
  }
}

static float spacing(float x) { return x < 1.0f ? x / 2.0f : 1.0f - 1.0f / (2.0f * x); }
static float spacing_inv(float x) { return x < 0.5f ? 2.0f * x : 1.0f / (2.0f - 2.0f * x); }

int main(int argc, char** argv) {
  const int N = 4096, S = argc > 1 ? atoi(argv[1]) : 48, L = 16, log2T = 19;
  const int64_t P = (int64_t)N * S;
  const uint32_t tsize = 1u << log2T;
  Res R;
  for (int l = 0; l < L; ++l) R.r[l] = floorf(16.0f * expf(l * logf(2048.0f / 16.0f) / (L - 1)));
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::normal_distribution<float> G(0.f, 1.f);
  std::vector<float> pos(P * 3);
  float o[3], d[3];
  for (int r = 0; r < N; ++r) {
    if ((r & 3) == 0) {
      float n = 0; for (int k = 0; k < 3; ++k) { o[k] = G(rng); n += o[k] * o[k]; }
      n = sqrtf(n); for (int k = 0; k < 3; ++k) o[k] = o[k] / n * (0.8f + 0.4f * U(rng));
      float t[3], m = 0; for (int k = 0; k < 3; ++k) { t[k] = 0.3f * G(rng) - o[k]; m += t[k] * t[k]; }
      m = sqrtf(m); for (int k = 0; k < 3; ++k) d[k] = t[k] / m;
    } else {
      for (int k = 0; k < 3; ++k) d[k] += 1e-3f * G(rng);  // neighbouring pixel of the patch
    }
    std::vector<float> x(S);
    for (auto& v : x) v = U(rng);
    std::sort(x.begin(), x.end());
    const float sn = spacing(0.05f), sf = spacing(1000.0f);
    for (int s = 0; s < S; ++s) {
      const float t = spacing_inv(x[s] * sf + (1 - x[s]) * sn);
      float w[3], mag = 0; for (int k = 0; k < 3; ++k) { w[k] = o[k] + d[k] * t; mag = fmaxf(mag, fabsf(w[k])); }
      for (int k = 0; k < 3; ++k) {
        float v = w[k]; if (!(mag < 1.0f)) v = (2.0f - 1.0f / mag) * (v / mag);
        v = (v + 2.0f) / 4.0f; pos[((int64_t)r * S + s) * 3 + k] = fminf(fmaxf(v, 1e-6f), 1.0f - 1e-6f);
      }
    }
  }
  float* dpos; float2* table; float4* encA; float2* encB;
  (void)hipMalloc(&dpos, P * 12); (void)hipMemcpy(dpos, pos.data(), P * 12, hipMemcpyHostToDevice);
  (void)hipMalloc(&table, (size_t)L * tsize * 8); (void)hipMemset(table, 0, (size_t)L * tsize * 8);
  (void)hipMalloc(&encA, ((P + 31) / 32) * 32 * 32 * 4); (void)hipMalloc(&encB, P * L * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    launch();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %7.1f us\n", name, ms * 100);
  };
  printf("P = %lld samples, %d levels x 2^%d\n", (long long)P, L, log2T);
  for (int grid : {512, 1024, 2048})
    timeit(("A fused mapping, grid " + std::to_string(grid)).c_str(), [&] { hipLaunchKernelGGL(k_fused_map, dim3(grid), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, encA); });
  {
    const int chunks = (int)((P + 255) / 256);
    timeit("B xcd-affine, 1 sample/thread", [&] { hipLaunchKernelGGL(k_xcd_map<1>, dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, encB); });
  }
  {
    const int chunks = (int)((P + 511) / 512);
    timeit("B xcd-affine, 2 samples/thread", [&] { hipLaunchKernelGGL(k_xcd_map<2>, dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, encB); });
  }
  {
    const int chunks = (int)((P + 1023) / 1024);
    timeit("B xcd-affine, 4 samples/thread", [&] { hipLaunchKernelGGL(k_xcd_map<4>, dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, encB); });
  }
  {
    const int chunks = (int)((P + 2047) / 2048);
    timeit("B xcd-affine, 8 samples/thread", [&] { hipLaunchKernelGGL(k_xcd_map<8>, dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, encB); });
  }
  {
    const int chunks = (int)((P + 511) / 512);
    timeit("C xcd-affine lane pairs, 4 samples/pair", [&] { hipLaunchKernelGGL(k_xcd_pair<4>, dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, (float*)encB); });
  }
  {
    const int chunks = (int)((P + 511) / 512);
    timeit("C lane pairs, nontemporal loads", [&] { hipLaunchKernelGGL((k_xcd_pair<4, 1>), dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, (float*)encB); });
    timeit("C lane pairs, agent-scope atomic loads", [&] { hipLaunchKernelGGL((k_xcd_pair<4, 2>), dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, (float*)encB); });
    timeit("C lane pairs, system-scope atomic loads", [&] { hipLaunchKernelGGL((k_xcd_pair<4, 3>), dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, (float*)encB); });
  }
  {
    const int chunks = (int)((P + 1023) / 1024);
    timeit("C xcd-affine lane pairs, 8 samples/pair", [&] { hipLaunchKernelGGL(k_xcd_pair<8>, dim3(8 * 2 * chunks), dim3(256), 0, 0, table, dpos, P, R, tsize - 1, tsize, chunks, (float*)encB); });
  }
  return 0;
}
