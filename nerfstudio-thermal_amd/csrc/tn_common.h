// Shared host/device helpers for libthermal_nerf_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/thermal_nerf_hip.h"

#define TN_WAVE 64

// ---------------------------------------------------------------- host side
void tn_set_error(const char* fmt, ...);
#define TN_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      tn_set_error(__VA_ARGS__);     \
      return TN_EINVAL;              \
    }                                \
  } while (0)
#define TN_CHECK_LAUNCH(name)                                         \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      tn_set_error("%s: %s", name, hipGetErrorString(e__));           \
      return TN_ELAUNCH;                                              \
    }                                                                 \
  } while (0)

static inline hipStream_t tn_s(tn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Fork/join inside one entry point: the weight-gradient GEMMs (HBM-stream + MFMA bound) and the table-gradient scatter (atomic-request
// bound) of a backward pass are independent and use different parts of the chip, so they run side by side.  tn_fork() makes the
// library-owned companion stream of `user` wait for everything enqueued on `user` so far and returns it; tn_join() makes `user` wait for
// the companion.  Both are plain event record/wait pairs (capturable into a hipGraph).  Companion streams are created once per
// (device, user stream) and live for the process.  Returns nullptr (caller then stays on `user`) if a stream/event cannot be created.
hipStream_t tn_fork(hipStream_t user);
// further companions of the same caller stream (idx 1, 2: the proposal networks' backward inside tn_render_rays_train_bwd); idx 0 == tn_fork
hipStream_t tn_fork_n(hipStream_t user, int idx);
void tn_join_n(hipStream_t user, int idx);
void tn_join(hipStream_t user, hipStream_t companion);
// join whatever companion stream `user` has (no-op if it never forked)
void tn_join_all(hipStream_t user);
__host__ __device__ static inline int64_t tn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// POD copy of TnGrid that is passed to kernels by value
struct GridK {
  const float2* table;
  float2* grad;
  int L;
  uint32_t mask;    // 2^log2T - 1
  uint32_t tsize;   // 2^log2T
  float res[TN_MAX_LEVELS];
  float* nonfinite;  // TnGrid::nonfinite_flag
  int grad_zero;     // TnGrid::table_grad_is_zero
};
static inline GridK make_gridk(const TnGrid& g) {
  GridK k;
  k.table = reinterpret_cast<const float2*>(g.table);
  k.grad = reinterpret_cast<float2*>(g.table_grad);
  k.L = g.num_levels;
  k.tsize = 1u << g.log2_hashmap_size;
  k.mask = k.tsize - 1u;
  for (int i = 0; i < TN_MAX_LEVELS; ++i) k.res[i] = g.res[i];
  k.nonfinite = g.nonfinite_flag;
  k.grad_zero = g.table_grad_is_zero;
  return k;
}

// tn_field_fwd with tn_field_pack_weights (pack_first) and a zero-fill of `zero` (16-byte aligned, zero_bytes a multiple of 16; or NULL) riding
// in its first launch: what tn_render_rays_eval / _train call (tn_field.hip)
int tn_prop_density_fwd_ex(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins, int64_t N, int32_t S,
                           float* density, float* enc_out, tn_stream_t stream);
int tn_prop_density_bwd_ex(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins, const float* d_density, int64_t N,
                           int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins, float* d_directions, const float* saved_enc,
                           tn_stream_t stream);
int tn_field_fwd_ex(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins, int64_t N, int32_t S,
                    int32_t training, void* workspace, int64_t workspace_bytes, float* density, float* rgb, float* density_pre, int pack_first, void* zero,
                    int64_t zero_bytes, tn_stream_t stream);

// ---------------------------------------------------------------- device side
#define TN_PRIME_Y 2654435761u
#define TN_PRIME_Z 805459861u

// torch.nan_to_num defaults: nan -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX
__device__ __forceinline__ float tn_nan_to_num(float x) {
  if (x != x) return 0.0f;
  if (x > 3.4028234663852886e38f) return 3.4028234663852886e38f;
  if (x < -3.4028234663852886e38f) return -3.4028234663852886e38f;
  return x;
}

// UniformLinDispPiecewiseSampler spacing function and inverse (model_components/ray_samplers.py:244-245)
__device__ __forceinline__ float tn_spacing(float x) { return x < 1.0f ? x / 2.0f : 1.0f - 1.0f / (2.0f * x); }
__device__ __forceinline__ float tn_spacing_inv(float x) { return x < 0.5f ? 2.0f * x : 1.0f / (2.0f - 2.0f * x); }
// spacing_to_euclidean_fn (model_components/ray_samplers.py:113-118): s_inv(x*s_far + (1-x)*s_near)
__device__ __forceinline__ float tn_s_to_euclid(float x, float s_near, float s_far) {
  return tn_spacing_inv(x * s_far + (1.0f - x) * s_near);
}

// SpacedSampler / UniformLinDispPiecewiseSampler bins of every ray (model_components/ray_samplers.py:78-128,225-248) as a grid-stride body
// i / d for a flat work-item index: a 64-bit division is ~150 instructions on this machine, a 32-bit one ~25.  `total` (wave-uniform) bounds i.
__device__ __forceinline__ int64_t tn_div_index(int64_t i, int64_t d, int64_t total) {
  return total < (1ll << 31) ? (int64_t)((uint32_t)i / (uint32_t)d) : i / d;
}

// (bid / nblk stand in for blockIdx.x / gridDim.x: tn_spaced_bins launches it alone, tn_pose_spaced_bins as one slice of a launch)
// One WAVE per ray (bin j = lane + 64 k): the ray's near / far / jitter and their spacings once per ray instead of once per bin (three of the four
// divisions of a bin were the ray's), consecutive lanes write consecutive bins.  Same expressions per bin as the reference's.
// the bins of ONE ray by its wave; e_keep (LDS, S + 1 floats) / s_keep (registers, bin j = lane + 64 k) : optional copies for stages that follow in the
// same wave (tn_next_sampling.h)
__device__ __forceinline__ void tn_spaced_bins_ray(const float* __restrict__ lin_bins, bool jittered, float jit, float s_near, float s_far, int S,
                                                   float* __restrict__ sb, float* __restrict__ eb, int lane, float* e_keep = nullptr,
                                                   float* s_keep = nullptr) {
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) {
    const int j = lane + 64 * k;
    if (j > S) continue;
    float b = lin_bins[j];
    if (jittered) {
      // bin_centers = (bins[1:]+bins[:-1])/2 ; upper = cat(centers, last) ; lower = cat(first, centers)
      float lower = (j == 0) ? lin_bins[0] : (lin_bins[j] + lin_bins[j - 1]) / 2.0f;
      float upper = (j == S) ? lin_bins[S] : (lin_bins[j + 1] + lin_bins[j]) / 2.0f;
      b = lower + (upper - lower) * jit;
    }
    const float e = tn_s_to_euclid(b, s_near, s_far);
    sb[j] = b;
    eb[j] = e;
    if (e_keep != nullptr) e_keep[j] = e;
    if (s_keep != nullptr) s_keep[k] = b;
  }
}
__device__ __forceinline__ void tn_spaced_bins_body(const float* __restrict__ lin_bins, const float* __restrict__ jitter,
                                                    const float* __restrict__ nears, const float* __restrict__ fars, int64_t N, int S,
                                                    float* __restrict__ s_bins, float* __restrict__ e_bins, int bid, int nblk) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  for (int64_t ray = (int64_t)bid * wpb + (threadIdx.x >> 6); ray < N; ray += (int64_t)nblk * wpb) {
    const float s_near = tn_spacing(nears[ray]), s_far = tn_spacing(fars[ray]);
    const float jit = jitter != nullptr ? jitter[ray] : 0.0f;
    tn_spaced_bins_ray(lin_bins, jitter != nullptr, jit, s_near, s_far, S, s_bins + ray * (S + 1), e_bins + ray * (S + 1), lane);
  }
}

// Frustums.get_positions + SceneContraction(L_inf) + (x+2)/4 + selector (cameras/rays.py:49-58,
// field_components/spatial_distortions.py:66-69, fields/density_fields.py:96-103).
// Returns the selector; p = masked unit-cube position; if jac != nullptr also d(p_unmasked)/d(world) facts for backward.
struct Contracted {
  float px, py, pz;  // masked [0,1]^3 position
  bool sel;
  // backward facts
  float wx, wy, wz;  // world position
  float mag;         // L_inf norm
  int amax;          // arg max |component|
};
__device__ __forceinline__ Contracted tn_contract(float ox, float oy, float oz, float dx, float dy, float dz, float start, float end) {
  Contracted c;
  // origins + directions * (starts + ends) / 2   (evaluated left to right as torch does)
  float se = start + end;
  c.wx = ox + (dx * se) / 2.0f;
  c.wy = oy + (dy * se) / 2.0f;
  c.wz = oz + (dz * se) / 2.0f;
  float ax = fabsf(c.wx), ay = fabsf(c.wy), az = fabsf(c.wz);
  float mag = fmaxf(fmaxf(ax, ay), az);
  c.mag = mag;
  c.amax = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
  float x = c.wx, y = c.wy, z = c.wz;
  if (!(mag < 1.0f)) {
    float k = 2.0f - (1.0f / mag);
    x = k * (x / mag);
    y = k * (y / mag);
    z = k * (z / mag);
  }
  x = (x + 2.0f) / 4.0f;
  y = (y + 2.0f) / 4.0f;
  z = (z + 2.0f) / 4.0f;
  c.sel = (x > 0.0f) && (x < 1.0f) && (y > 0.0f) && (y < 1.0f) && (z > 0.0f) && (z < 1.0f);
  float m = c.sel ? 1.0f : 0.0f;
  c.px = x * m;
  c.py = y * m;
  c.pz = z * m;
  return c;
}

// Backward of tn_contract: gradient wrt the masked unit-cube position -> gradient wrt the world position.
__device__ __forceinline__ void tn_contract_bwd(const Contracted& c, float gx, float gy, float gz, float& wx, float& wy, float& wz) {
  if (!c.sel) { wx = wy = wz = 0.0f; return; }
  gx *= 0.25f; gy *= 0.25f; gz *= 0.25f;  // (x+2)/4
  if (c.mag < 1.0f) { wx = gx; wy = gy; wz = gz; return; }
  // y_i = (2 - 1/m) x_i / m = (2/m - 1/m^2) x_i,  m = |x_a|  (a = argmax)
  // dy_i/dx_j = (2/m - 1/m^2) delta_ij + x_i * (-2/m^2 + 2/m^3) * sign(x_a) * delta_ja
  float m = c.mag;
  float s = 2.0f / m - 1.0f / (m * m);
  float t = (-2.0f / (m * m) + 2.0f / (m * m * m));
  float dot = gx * c.wx + gy * c.wy + gz * c.wz;
  wx = s * gx; wy = s * gy; wz = s * gz;
  float xa = c.amax == 0 ? c.wx : (c.amax == 1 ? c.wy : c.wz);
  float extra = dot * t * (xa >= 0.0f ? 1.0f : -1.0f);
  if (c.amax == 0) wx += extra; else if (c.amax == 1) wy += extra; else wz += extra;
}

// One level of the reference's torch hash encoding (field_components/encodings.py:420-461):
// ceil/floor corners, every level hashed, x-then-y-then-z interpolation in this exact operation order.
struct LevelCorners {
  uint32_t idx[8];  // f0..f7 in the reference's naming
  float ox, oy, oz;
};
__device__ __forceinline__ void tn_level_corners(float px, float py, float pz, float res, uint32_t mask, uint32_t level_off, LevelCorners& lc) {
  float sx = px * res, sy = py * res, sz = pz * res;
  float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
  uint32_t cx = (uint32_t)(int)ceilf(sx), cy = (uint32_t)(int)ceilf(sy), cz = (uint32_t)(int)ceilf(sz);
  uint32_t fx = (uint32_t)(int)fxf, fy = (uint32_t)(int)fyf, fz = (uint32_t)(int)fzf;
  lc.ox = sx - fxf; lc.oy = sy - fyf; lc.oz = sz - fzf;
  uint32_t hcy = cy * TN_PRIME_Y, hfy = fy * TN_PRIME_Y, hcz = cz * TN_PRIME_Z, hfz = fz * TN_PRIME_Z;
  lc.idx[0] = ((cx ^ hcy ^ hcz) & mask) + level_off;  // (c,c,c)
  lc.idx[1] = ((cx ^ hfy ^ hcz) & mask) + level_off;  // (c,f,c)
  lc.idx[2] = ((fx ^ hfy ^ hcz) & mask) + level_off;  // (f,f,c)
  lc.idx[3] = ((fx ^ hcy ^ hcz) & mask) + level_off;  // (f,c,c)
  lc.idx[4] = ((cx ^ hcy ^ hfz) & mask) + level_off;  // (c,c,f)
  lc.idx[5] = ((cx ^ hfy ^ hfz) & mask) + level_off;  // (c,f,f)
  lc.idx[6] = ((fx ^ hfy ^ hfz) & mask) + level_off;  // (f,f,f)
  lc.idx[7] = ((fx ^ hcy ^ hfz) & mask) + level_off;  // (f,c,f)
}
// The two features of a level are interpolated TOGETHER, as 2-vectors: the same multiplies and adds per component in the same order (IEEE, no
// contraction: bit-identical to the scalar form), issued as packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32: two floats per lane and
// instruction) -- the interpolation is ~210 of the ~930 instructions a proposal sample costs, and those kernels are bound by instruction issue.
typedef float tn_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 tn_level_interp(const float2 f[8], float ox, float oy, float oz) {
  const float ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
  tn_v2f v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i].x = f[i].x; v[i].y = f[i].y; }
  const tn_v2f f03 = v[0] * ox + v[3] * ux;
  const tn_v2f f12 = v[1] * ox + v[2] * ux;
  const tn_v2f f56 = v[5] * ox + v[6] * ux;
  const tn_v2f f47 = v[4] * ox + v[7] * ux;
  const tn_v2f f0312 = f03 * oy + f12 * uy;
  const tn_v2f f4756 = f47 * oy + f56 * uy;
  const tn_v2f r = f0312 * oz + f4756 * uz;
  return make_float2(r.x, r.y);
}
// Trilinear value AND its derivatives wrt the in-cell offset (o = scaled - floor(scaled); d o / d position = res), feature by feature:
// what autograd derives from tn_level_interp's operation sequence (field_components/encodings.py:449-459 with offset = scaled - scaled_f).
// jac[3 f + axis] = res * d enc_f / d o_axis
__device__ __forceinline__ float2 tn_level_interp_jac(const float2 f[8], float ox, float oy, float oz, float res, float jac[6]) {
  float ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
  float2 r;
#define TN_LERP3J(C, K)                                                         \
  {                                                                              \
    float f03 = f[0].C * ox + f[3].C * ux;                                       \
    float f12 = f[1].C * ox + f[2].C * ux;                                       \
    float f56 = f[5].C * ox + f[6].C * ux;                                       \
    float f47 = f[4].C * ox + f[7].C * ux;                                       \
    float f0312 = f03 * oy + f12 * uy;                                           \
    float f4756 = f47 * oy + f56 * uy;                                           \
    r.C = f0312 * oz + f4756 * uz;                                               \
    float dx = ((f[0].C - f[3].C) * oy + (f[1].C - f[2].C) * uy) * oz + ((f[4].C - f[7].C) * oy + (f[5].C - f[6].C) * uy) * uz; \
    float dy = (f03 - f12) * oz + (f47 - f56) * uz;                              \
    float dz = f0312 - f4756;                                                    \
    jac[3 * K] = dx * res; jac[3 * K + 1] = dy * res; jac[3 * K + 2] = dz * res; \
  }
  TN_LERP3J(x, 0)
  TN_LERP3J(y, 1)
#undef TN_LERP3J
  return r;
}
// (a table has at most 16 levels x 2^24 entries of 8 bytes: the byte offset of an entry fits 32 bits, and a 32-bit offset on a wave-uniform base is
// ONE address instruction per gather instead of a 64-bit shift-and-add)
__device__ __forceinline__ float2 tn_table_entry(const float2* __restrict__ table, uint32_t idx) {
  return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(table) + (uint32_t)(idx * 8u));
}
__device__ __forceinline__ float2 tn_encode_level_jac(const float2* __restrict__ table, float px, float py, float pz, float res, uint32_t mask,
                                                      uint32_t level_off, float jac[6]) {
  LevelCorners lc;
  tn_level_corners(px, py, pz, res, mask, level_off, lc);
  float2 f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = tn_table_entry(table, lc.idx[i]);
  return tn_level_interp_jac(f, lc.ox, lc.oy, lc.oz, res, jac);
}
__device__ __forceinline__ float2 tn_encode_level(const float2* __restrict__ table, float px, float py, float pz, float res, uint32_t mask,
                                                  uint32_t level_off) {
  LevelCorners lc;
  tn_level_corners(px, py, pz, res, mask, level_off, lc);
  float2 f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = tn_table_entry(table, lc.idx[i]);
  return tn_level_interp(f, lc.ox, lc.oy, lc.oz);
}

// ---------------------------------------------------------------- wave primitives (64 lanes)
__device__ __forceinline__ int tn_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ float tn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double tn_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float tn_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float tn_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive prefix sum across the wave (double: torch's CPU cumsum accumulates float inputs in double)
__device__ __forceinline__ double tn_wave_incl_scan_d(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// exclusive prefix over earlier lanes from an inclusive scan, WITHOUT a subtraction (incl - own leaves cancellation residue where the
// true prefix is exactly 0, and Adam with eps=1e-15 turns a 1e-16 gradient residue into a full-size parameter step)
__device__ __forceinline__ double tn_excl_from_incl_d(double incl, int lane) {
  double t = __shfl_up(incl, 1, 64);
  return lane == 0 ? 0.0 : t;
}
__device__ __forceinline__ float tn_rexcl_from_incl(float incl, int lane) {
  float t = __shfl_down(incl, 1, 64);
  return lane == 63 ? 0.0f : t;
}
// inclusive suffix sum across the wave (float)
__device__ __forceinline__ float tn_wave_incl_rscan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_down(v, o, 64);
    if (lane + o < 64) v += t;
  }
  return v;
}

// Work-item -> (ray, sample) in PATCH order: groups of 4 consecutive rays (a 2x2 pixel patch from PatchPixelSampler: neighbouring pixels,
// near-identical rays) walked sample-major, so consecutive lanes are the same depth of neighbouring rays, then the next depth.  Runs of
// lanes in one grid cell become ~3x longer than in ray-major order (measured in round 1), which is what the run-length merging of the
// scatter-add feeds on.  Any N is handled (the last group may hold fewer than 4 rays).
__device__ __forceinline__ void tn_patch_order(int64_t i, int64_t N, int S, int64_t& ray, int& s) {
  int64_t per = 4 * (int64_t)S;
  int64_t g = tn_div_index(i, per, N * (int64_t)S);
  int w = (int)(i - g * per);
  int64_t r0 = g * 4;
  int nr = (N - r0) < 4 ? (int)(N - r0) : 4;
  s = nr == 4 ? (w >> 2) : w / nr;
  ray = r0 + (w - s * nr);
}

// ---------------------------------------------------------------- the shared table-gradient scatter kernel (tn_scatter.hip)
// Coarse levels touch only (res+1)^3 table entries and the scene contraction concentrates the samples on a few of them: atomics to one
// 64-B line serialise (scripts/microbench/atomic_hotset.hip: 64 hot lines -> 3.5 G requests/s instead of 21 G), so a level-0 launch
// is 2x slower than a fine level with 8x the requests.  Such levels are accumulated into R private replicas (a wave adds into replica
// (wave-id) % R: the waves of a block walk adjacent samples) in the dense layout index (z*r1 + y)*r1 + x, r1 = res + 1, and
// k_replica_reduce folds the replicas into the hashed gradient afterwards.
#define TN_SCATTER_SCRATCH_BYTES (64ll << 20)  // replica scratch of the atomic path (part of every backward workspace)
// The default path is atomic-free ("binned", tn_scatter.hip): contributions are written once as (slot, value) records into buckets of
// 2^TN_BIN_SLICE_LOG2 consecutive table slots and summed per bucket in LDS.  Scratch per level: room for 16 records per sample (twice the
// 8 corners: the hash spreads the records evenly, coarse levels are merged before they are written) plus slack for small batches.
#ifndef TN_BIN_SLICE_LOG2
#define TN_BIN_SLICE_LOG2 12
#endif
#define TN_BIN_MAX_SLICES 256
#define TN_BIN_COUNT_STRIDE 16  // words reserved per bucket counter (one 64-B line each)
static inline int64_t tn_bin_level_records(int64_t P) { return 16 * P + TN_BIN_MAX_SLICES * 1032; }
static inline int64_t tn_bin_bytes(int64_t P, int num_levels) {
  return 256 + (int64_t)num_levels * TN_BIN_MAX_SLICES * 4 * TN_BIN_COUNT_STRIDE + (int64_t)num_levels * tn_bin_level_records(P) * (2 + 8);
}
// bytes of scatter scratch a backward workspace carries for P samples on a grid of num_levels levels
static inline int64_t tn_scatter_scratch_bytes(int64_t P, int num_levels) {
  int64_t b = tn_bin_bytes(P, num_levels);
  return ((b > TN_SCATTER_SCRATCH_BYTES ? b : TN_SCATTER_SCRATCH_BYTES) + 255) / 256 * 256;
}
enum { TN_REP_NONE = 0, TN_REP_DENSE = 1 };
struct ReplicaK {
  float2* rep;                     // scratch, zero-filled before the scatter
  uint32_t total;                  // sum of n[l] over the replicated levels (= threads of the reduce kernel)
  uint32_t kinds;                  // 2 bits per level: TN_REP_*
  uint32_t n[TN_MAX_LEVELS];       // entries of one replica of level l
  uint32_t off[TN_MAX_LEVELS];     // scratch offset (entries) of replica 0 of level l; replica r is at off + r * n
  uint32_t first[TN_MAX_LEVELS];   // prefix sum of n over the replicated levels (reduce-thread index of the level's entry 0)
  uint8_t R[TN_MAX_LEVELS];        // replicas of level l
};
// g_enc: [P, ld] gradient of the encoding (feature 2*level + f), rows in ray-major sample order; or, with ld == TN_LD_LEVEL_MAJOR,
// [levels][P] float2 (level-major: what the main field's backward writes for the bin pass and k_field_dpos).
#define TN_LD_LEVEL_MAJOR (-1)
// scratch: tn_scatter_scratch_bytes(N*S, levels) of device memory or NULL (every level then adds straight into the hashed gradient).
// dense_sum: NULL, or [tn_grid_dense_count(grid, N*S)] float2 that receive the per-cell sums INSTEAD of the hashed gradient (every level
// of the grid must then be a dense-replica level); tn_grid_dense_fold adds such sums into the hashed gradient later.
// tn_adam_step_ranges_amp_update with the next iteration's tn_sample_rays in co-work blocks of the launch (next may be NULL) -- or, with `chain`,
// the next iteration's whole sampling front for that batch (tn_next_sampling.h; needs `next`).  su_update = false: no GradScaler.update() in this
// launch (a launch that is not the iteration's last optimiser launch).
struct TnNextSamplingHost {
  const TnPropNet *prop0, *prop1;
  const float* pose; const uint8_t* frozen; int num_cameras;
  const float *rays_o, *rays_d; const int64_t* cam;  // the batch (an earlier launch has sampled it)
  const float *nears, *fars, *jit0, *jit1, *jit2, *lin0, *lin1, *lin2;
  float anneal;
  int S0, S1, S2;
  int64_t N;
  float* out;       // the NEXT forward buffer
  int64_t off[16];  // float offsets of its regions: origins, directions | s0 e0 d0 w0 m0 | s1 e1 d1 w1 m1 | s2 e2 | penc0 penc1
  int save_enc;     // keep the proposal levels' encodings
};
int tn_adam_step_ranges_amp_update_cw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                                      const int64_t* counts, const int32_t* steps, const double* lrs, const double* lr_finals,
                                      const int32_t* sched_max_steps, int32_t sched_step, double beta1, double beta2, double eps, const float* inv_scale,
                                      float* found_inf, const int32_t* flag_index, int32_t num_flags, int32_t* skipped, int32_t lag_index,
                                      int32_t count_skip, int32_t zero_grads, float* scale, int32_t* growth_tracker, uint32_t* done_counter,
                                      double growth_factor, double backoff_factor, int32_t growth_interval, const TnSampleRays* next, bool* next_taken,
                                      tn_stream_t stream, const TnNextSamplingHost* chain = nullptr, bool su_update = true);
// whether the chain supports these sample counts (the lane layouts of the default sampler: 4 / 2 samples per lane on the proposal levels)
static inline bool tn_next_sampling_supported(int S0, int S1, int S2) { return S0 > 128 && S0 <= 256 && S1 > 64 && S1 <= 128 && S2 >= 1 && S2 <= 256; }
// internal variants of tn_render_fwd / tn_train_losses for tn_train_step: the batch-wide clip of depth_expected not as a launch of its own behind
// the renderers but as co-work blocks of the loss launch that follows (tn_sampler.hip)
int tn_render_fwd_ex(const float* e_bins, const float* density, const float* rgb, int64_t N, int32_t S, int32_t C, int32_t training, float* weights,
                     float* comp, float* accumulation, float* depth_median, float* depth_expected, float* scratch, bool launch_clip, int* clip_nblk,
                     tn_stream_t stream);
int tn_train_losses_clip(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, int32_t num_props, const float* const* s_bins_prop,
                         const float* const* weights_prop, const int32_t* S_prop, float* const* d_weights_prop, int64_t N, float distortion_mult,
                         float interlevel_mult, float* d_weights_fine, const float* pred_rgb, int32_t rgb_stride, const float* pred_thermal,
                         int32_t thermal_stride, const float* image, const float* is_thermal, float thermal_mult, float tv_mult, float cross_mult,
                         float* d_pred_rgb, float* d_pred_thermal, float* loss_lines, float* depth_expected, const float* scratch, int clip_nblk,
                         tn_stream_t stream);
// cowork: NULL, or the main field's d position pass (tn_field_dpos.h) to run in extra blocks of the bin launch -- only where
// tn_grid_scatter_takes_cowork says so (the segmented path, no d position of the scatter's own)
// fold_cowork: NULL, or the launch that ends an iteration's backward (tn_pose_finish.h) to run in the first blocks of the FOLD launch (same condition)
struct DposArgs;
struct PoseFinishArgs;
int tn_grid_scatter_launch(const TnGrid& grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int ld,
                           int64_t N, int S, float* d_origins, float* d_directions, void* scratch, hipStream_t stream, float* dense_sum = nullptr,
                           bool counters_zeroed = false, const DposArgs* cowork = nullptr, const PoseFinishArgs* fold_cowork = nullptr);
bool tn_grid_scatter_takes_cowork(const TnGrid& grid, int64_t P, void* scratch);
// tn_field_bwd_phase with the fold's co-work (tn_train_step); *fold_cowork_taken says whether the fold launch carried it (else the caller launches it)
int tn_field_bwd_phase_ex(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins,
                          const float* d_density, const float* d_rgb, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins,
                          float* d_directions, int32_t phases, int32_t level_begin, int32_t level_end, const PoseFinishArgs* fold_cowork,
                          bool* fold_cowork_taken, tn_stream_t stream);
int tn_pose_finish_args(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                        const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose, const float* loss_lines,
                        float* losses16, float trans_pen, float rot_pen, float scale, float* reg_out, const float* grads, int32_t num_ranges,
                        const int64_t* offsets, const int64_t* counts, const int32_t* flag_index, int32_t num_flags, float* found_inf, int32_t pose_flag,
                        PoseFinishArgs& a);
// The bin pass needs its bucket counters zero.  A separate hipMemsetAsync of those few KB costs 6-25 us on the launch stream (config 2 makes
// seven of them per step): the kernel that produces d enc -- always the launch right before the scatter on the same stream -- zeroes them instead
// (tn_zero_words, first thing block 0 does) and the scatter is told so (counters_zeroed).  *words = 0 when the binned path is not taken.
void tn_grid_scatter_counters(const TnGrid& grid, int64_t P, void* scratch, uint32_t** ptr, int* words);
__device__ __forceinline__ void tn_zero_words(uint32_t* __restrict__ p, int words) {
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < words; i += blockDim.x) p[i] = 0u;
}
// The binned scatter in two steps, for a backward that hands level ranges to a gradient exchange as they complete: the bin pass covers the
// whole grid once, the fold runs per level range.  Only valid when tn_grid_scatter_is_binned(grid, P, scratch).
bool tn_grid_scatter_is_binned(const TnGrid& grid, int64_t P, const void* scratch);
int tn_grid_scatter_bin(const TnGrid& grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int ld, int64_t N,
                        int S, float* d_origins, float* d_directions, void* scratch, hipStream_t stream, bool counters_zeroed = false,
                        const DposArgs* cowork = nullptr);
int tn_grid_scatter_fold(const TnGrid& grid, int64_t P, void* scratch, int level_begin, int level_end, hipStream_t stream,
                         const PoseFinishArgs* cowork = nullptr);
int64_t tn_grid_dense_count(const TnGrid& grid, int64_t P);
int tn_grid_dense_fold(const TnGrid& grid, int64_t P, const float* dense_sum, hipStream_t stream);
