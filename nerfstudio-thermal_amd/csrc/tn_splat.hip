// N4: forward Gaussian-splat render with an RGB + thermal colour per Gaussian, for gfx950.
//
// What it replaces: the three gsplat calls of SplatfactoModel.get_outputs (nerfstudio/models/splatfacto.py:739-807: project_gaussians,
// spherical_harmonics, rasterize_gaussians twice -- colour and depth).  gsplat (>=0.1.6, pyproject.toml:66) is a third-party CUDA package
// that is not in the reference tree: the arithmetic below follows its published algorithm (see oracle/splat_oracle.py; parity unpinned).
//
// Pipeline (4 launches + one radix sort):
//   k_splat_project    thread = Gaussian: view transform, 3D covariance, EWA projection (+0.3 px blur), conic, 3-sigma radius, tile bounding
//                      box; for visible Gaussians also the view-dependent colour (degree <= 3 SH, RGB and thermal) and the opacity -> one
//                      packed 48-byte record the rasteriser reads with three 16-byte loads
//   depth sort         rocprim radix sort of the N Gaussians by depth bits (32-bit keys, N elements)
//   rocprim scan       inclusive scan of tiles-per-Gaussian in depth order
//   k_splat_intersect  one (tile id, Gaussian id) pair per (Gaussian, tile), emitted in depth order; a wave owns 64 Gaussians and writes
//                      their pairs as ONE contiguous, coalesced range (lane = output position, owner found by binary search in LDS)
//   rocprim radix sort of the pairs on log2(#tiles) bits only (13 bits at 1080p = 2 passes over 8-byte pairs).  gsplat sorts
//                      (tile << 32 | depth) 64-bit keys: 6 passes over 12-byte pairs, 46 % of the frame in the first version of this
//                      file.  A stable sort by tile of a depth-ordered list gives the same order, ties included (equal depths keep the
//                      Gaussian order in both).
//   k_splat_tile_edges start / end of every tile's run
//   k_splat_raster     block = one 16x16 tile = 4 waves, lane = pixel.  256 records at a time are staged through LDS (each thread fetches
//                      one); every lane then walks the batch front to back.  The LDS reads are wave-uniform broadcasts (conflict-free);
//                      a wave whose 64 pixels are all finished skips the arithmetic; the colour (RGB+T) and depth images come out of ONE
//                      pass (the reference runs the rasteriser twice; in "antialiased" mode the depth pass uses the uncompensated
//                      opacity, so that variant carries a second transmittance).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "tn_common.h"

#define SPLAT_BLOCK 16
#define SPLAT_BATCH 256

// alpha = opacity exp(-sigma), sigma = 0.5 (cx dx^2 + cz dy^2) + cy dx dy, is evaluated as exp2(l2op - (A dx^2 + B dx dy + C dy^2)) with
// A = 0.5 cx log2(e), B = cy log2(e), C = 0.5 cz log2(e), l2op = log2(opacity): 5 FMAs + one v_exp_f32 per pixel instead of 9 multiply/adds,
// two scalings and a multiply by the opacity (the rasteriser is VALU-bound: rocprofv3 SQ_ACTIVE_INST_VALU = 70 % of its duration).
struct SplatRec {  // 64 bytes per Gaussian
  float4 a;        // x, y, A, B
  float4 b;        // C, l2op (compensated opacity in antialiased mode), hx, hy: half extents of the box around {alpha >= 1/255}
  float4 c;        // r, g, b, thermal
  float4 d;        // depth, l2op of the plain opacity, -, -
};
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SplatWs {
  SplatRec* recs;
  int32_t* cum;        // inclusive scan of the tight tile counts in depth order
  int32_t* tbox;       // [N][4] tight tile box (x0, y0, x1, y1): gsplat's 3-sigma box cut down to the tiles alpha >= 1/255 can reach
  int32_t* thits;      // [N] tiles in the tight box
  int32_t* tile_bins;  // [num_tiles][2]
  uint32_t* lkeys[2];  // ~run length per tile (sort key: longest first)
  int32_t* lvals[2];   // tile ids; lvals[1] = the order the rasteriser takes the tiles in
  uint32_t* depth_max; // float bits (depths are positive)
  uint32_t* dkeys;     // sorted depth bits [N]
  int32_t* order;      // Gaussian ids in depth order [N]
  uint32_t* keys[2];   // tile id per intersection
  int32_t* vals[2];    // Gaussian id per intersection
  void* tmp;
  size_t tmp_bytes;
};

struct HitsInOrder {  // tiles-per-Gaussian read through the depth order (input of the scan)
  const int32_t* hits;
  __host__ __device__ int32_t operator()(int32_t id) const { return hits[id]; }
};

static size_t al256(size_t x) { return (x + 255) / 256 * 256; }

static size_t sort_tmp_bytes(int64_t capacity, int64_t N) {
  size_t a = 0, b = 0, c = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)std::max<int64_t>(capacity, 1 << 20), 0, 32);  // also covers the sort of the tile order (<= 2^20 tiles)
  (void)rocprim::radix_sort_pairs(nullptr, c, (uint32_t*)nullptr, (uint32_t*)nullptr, rocprim::counting_iterator<int32_t>(0), (int32_t*)nullptr,
                                  (size_t)std::max<int64_t>(N, 1), 0, 32);
  (void)rocprim::inclusive_scan(nullptr, b, rocprim::make_transform_iterator((const int32_t*)nullptr, HitsInOrder{nullptr}), (int32_t*)nullptr,
                                (size_t)std::max<int64_t>(N, 1), rocprim::plus<int32_t>());
  return al256(std::max(std::max(a, b), c)) + 4096;
}

static SplatWs splat_layout(void* base, int64_t N, int64_t capacity, int32_t num_tiles, size_t* total) {
  SplatWs w;
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return base ? (void*)(p + o) : (void*)nullptr; };
  w.recs = (SplatRec*)take(sizeof(SplatRec) * (size_t)N);
  w.cum = (int32_t*)take(4 * (size_t)N);
  w.tbox = (int32_t*)take(16 * (size_t)N);
  w.thits = (int32_t*)take(4 * (size_t)N);
  w.tile_bins = (int32_t*)take(8 * (size_t)num_tiles);
  w.depth_max = (uint32_t*)take(256);
  for (int i = 0; i < 2; ++i) w.lkeys[i] = (uint32_t*)take(4 * (size_t)num_tiles);
  for (int i = 0; i < 2; ++i) w.lvals[i] = (int32_t*)take(4 * (size_t)num_tiles);
  w.dkeys = (uint32_t*)take(4 * (size_t)N);
  w.order = (int32_t*)take(4 * (size_t)N);
  for (int i = 0; i < 2; ++i) w.keys[i] = (uint32_t*)take(4 * (size_t)capacity);
  for (int i = 0; i < 2; ++i) w.vals[i] = (int32_t*)take(4 * (size_t)capacity);
  w.tmp_bytes = sort_tmp_bytes(capacity, N);
  w.tmp = take(w.tmp_bytes);
  if (total) *total = off;
  return w;
}

extern "C" int64_t tn_splat_workspace_bytes(int64_t num_gaussians, int64_t max_intersections, int32_t num_tiles) {
  if (num_gaussians < 0 || max_intersections < 0 || num_tiles < 1) return -1;
  size_t total = 0;
  (void)splat_layout(nullptr, num_gaussians, max_intersections, num_tiles, &total);
  return (int64_t)total;
}

// ------------------------------------------------------------------------------------------------ projection + colour
__device__ __forceinline__ float sh_eval(int degree, float x, float y, float z, const float* __restrict__ dc, const float* __restrict__ rest, int stride, int ch) {
  // coefficient k of channel ch: k == 0 -> dc[ch], else rest[(k-1)*stride + ch]
  float v = 0.28209479177387814f * dc[ch];
  if (degree < 1) return v;
#define CO(k) rest[((k) - 1) * stride + ch]
  v += 0.4886025119029199f * (-y * CO(1) + z * CO(2) - x * CO(3));
  if (degree < 2) return v;
  float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  v += 1.0925484305920792f * xy * CO(4) + -1.0925484305920792f * yz * CO(5) + 0.31539156525252005f * (2.0f * zz - xx - yy) * CO(6) +
       -1.0925484305920792f * xz * CO(7) + 0.5462742152960396f * (xx - yy) * CO(8);
  if (degree < 3) return v;
  v += -0.5900435899266435f * y * (3.0f * xx - yy) * CO(9) + 2.890611442640554f * xy * z * CO(10) + -0.4570457994644658f * y * (4.0f * zz - xx - yy) * CO(11) +
       0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * CO(12) + -0.4570457994644658f * x * (4.0f * zz - xx - yy) * CO(13) +
       1.445305721320277f * z * (xx - yy) * CO(14) + -0.5900435899266435f * x * (xx - 3.0f * yy) * CO(15);
#undef CO
  return v;
}

struct SplatCamK {
  float view[12];  // rows of the world->camera matrix (3x4)
  float proj[16];  // full projection matrix (4x4, row-major)
  float fx, fy, cx, cy, clip;
  float pos[3];
  int W, H, tbx, tby;
};

__global__ void __launch_bounds__(256) k_splat_project(SplatCamK cam, const float* __restrict__ means, const float* __restrict__ log_scales,
                                                       const float* __restrict__ quats, const float* __restrict__ opac_logit,
                                                       const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                                                       const float* __restrict__ t_dc, const float* __restrict__ t_rest, int64_t N, int sh_degree,
                                                       int rest_coeffs, int antialiased, float2* __restrict__ xys, float* __restrict__ depths,
                                                       int32_t* __restrict__ radii, float* __restrict__ conics, float* __restrict__ comp_out,
                                                       int32_t* __restrict__ tiles_hit, int32_t* __restrict__ tile_box, SplatRec* __restrict__ recs,
                                                       int32_t* __restrict__ tbox, int32_t* __restrict__ thits) {
  // The higher-order SH coefficients of the block's 256 Gaussians (45 + 15 floats each) go through LDS: the block copies its contiguous
  // 46 KB + 15 KB slab with coalesced 16-byte loads, and each thread then reads its own coefficients at stride 45 / 15 floats -- odd strides,
  // so the 64 lanes of a wave hit 64 different banks.  Reading them straight from global memory (each lane its own 180-byte run) cost
  // 3.6x the requests the data needs (TCC_REQ 17 M x 64 B for 300 MB) and made the kernel latency-bound (150 us per 1 M Gaussians; 124 us with the staging at 128 Gaussians per block).
  extern __shared__ float sh_lds[];
  float* s_rest = sh_lds;
  float* s_trest = sh_lds + (int)blockDim.x * rest_coeffs * 3;
  {
    const int64_t g0 = blockIdx.x * (int64_t)blockDim.x;
    const int cnt = (int)min((int64_t)blockDim.x, N - g0);
    if (rest_coeffs > 0 && sh_degree >= 1) {
      const int n3 = cnt * rest_coeffs * 3, n1 = cnt * rest_coeffs;
      const float* src3 = f_rest + g0 * rest_coeffs * 3;
      const float* src1 = t_rest + g0 * rest_coeffs;
      // both slabs start 16-byte aligned: blockDim * K * 3 * 4 and blockDim * K * 4 bytes per block are multiples of 16
      for (int t = threadIdx.x * 4; t + 3 < n3; t += blockDim.x * 4) *reinterpret_cast<float4*>(s_rest + t) = *reinterpret_cast<const float4*>(src3 + t);
      for (int t = (n3 & ~3) + threadIdx.x; t < n3; t += blockDim.x) s_rest[t] = src3[t];
      for (int t = threadIdx.x * 4; t + 3 < n1; t += blockDim.x * 4) *reinterpret_cast<float4*>(s_trest + t) = *reinterpret_cast<const float4*>(src1 + t);
      for (int t = (n1 & ~3) + threadIdx.x; t < n1; t += blockDim.x) s_trest[t] = src1[t];
    }
    __syncthreads();
  }
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  float2 xy = make_float2(0.f, 0.f);
  float depth = 0.f, cmp = 0.f;
  float3 conic = make_float3(0.f, 0.f, 0.f);
  int radius = 0, area = 0;
  int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
  const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
  const float* V = cam.view;
  float px = V[0] * mx + V[1] * my + V[2] * mz + V[3];
  float py = V[4] * mx + V[5] * my + V[6] * mz + V[7];
  float pz = V[8] * mx + V[9] * my + V[10] * mz + V[11];
  bool ok = pz > cam.clip;
  if (ok) {
    // Sigma = (R S)(R S)^T
    float qw = quats[4 * i], qx = quats[4 * i + 1], qy = quats[4 * i + 2], qz = quats[4 * i + 3];
    float qn = 1.0f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    qw *= qn; qx *= qn; qy *= qn; qz *= qn;
    float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy),
                  2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx),
                  2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy)};
    float s[3] = {expf(log_scales[3 * i]), expf(log_scales[3 * i + 1]), expf(log_scales[3 * i + 2])};
    float M[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) M[3 * r + c] = R[3 * r + c] * s[c];
    float S[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) S[3 * r + c] = M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1] + M[3 * r + 2] * M[3 * c + 2];
    // EWA: clamp to 1.3x the frustum, T = J W, cov2d = T Sigma T^T + 0.3 I
    float tan_x = 0.5f * (float)cam.W / cam.fx, tan_y = 0.5f * (float)cam.H / cam.fy;
    float lx = 1.3f * tan_x, ly = 1.3f * tan_y;
    float tx = pz * fminf(lx, fmaxf(-lx, px / pz));
    float ty = pz * fminf(ly, fmaxf(-ly, py / pz));
    float rz = 1.0f / pz, rz2 = rz * rz;
    float J0[3] = {cam.fx * rz, 0.f, -cam.fx * tx * rz2};
    float J1[3] = {0.f, cam.fy * rz, -cam.fy * ty * rz2};
    float T0[3], T1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      T0[c] = J0[0] * V[c] + J0[1] * V[4 + c] + J0[2] * V[8 + c];
      T1[c] = J1[0] * V[c] + J1[1] * V[4 + c] + J1[2] * V[8 + c];
    }
    float A0[3], A1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      A0[c] = T0[0] * S[c] + T0[1] * S[3 + c] + T0[2] * S[6 + c];
      A1[c] = T1[0] * S[c] + T1[1] * S[3 + c] + T1[2] * S[6 + c];
    }
    float c00 = A0[0] * T0[0] + A0[1] * T0[1] + A0[2] * T0[2];
    float c01 = A0[0] * T1[0] + A0[1] * T1[1] + A0[2] * T1[2];
    float c11 = A1[0] * T1[0] + A1[1] * T1[1] + A1[2] * T1[2];
    float det_orig = c00 * c11 - c01 * c01;
    float a = c00 + 0.3f, b = c01, c = c11 + 0.3f;
    float det = a * c - b * b;
    cmp = sqrtf(fmaxf(0.f, det_orig / det));
    ok = det != 0.f;
    if (ok) {
      float inv = 1.0f / det;
      conic = make_float3(c * inv, -b * inv, a * inv);
      float mid = 0.5f * (a + c);
      float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
      radius = (int)ceilf(3.0f * sqrtf(fmaxf(mid + disc, mid - disc)));
      const float* P = cam.proj;
      float hx = P[0] * mx + P[1] * my + P[2] * mz + P[3];
      float hy = P[4] * mx + P[5] * my + P[6] * mz + P[7];
      float hw = P[12] * mx + P[13] * my + P[14] * mz + P[15];
      float rw = 1.0f / (hw + 1e-6f);
      xy = make_float2(0.5f * (float)cam.W * (hx * rw) + cam.cx - 0.5f, 0.5f * (float)cam.H * (hy * rw) + cam.cy - 0.5f);
      float tcx = xy.x / (float)SPLAT_BLOCK, tcy = xy.y / (float)SPLAT_BLOCK, tr = (float)radius / (float)SPLAT_BLOCK;
      x0 = min(max(0, (int)(tcx - tr)), cam.tbx);
      x1 = min(max(0, (int)(tcx + tr + 1.f)), cam.tbx);
      y0 = min(max(0, (int)(tcy - tr)), cam.tby);
      y1 = min(max(0, (int)(tcy + tr + 1.f)), cam.tby);
      area = (x1 - x0) * (y1 - y0);
      ok = area > 0;
      depth = pz;
    }
  }
  if (!ok) {
    xy = make_float2(0.f, 0.f); depth = 0.f; conic = make_float3(0.f, 0.f, 0.f); radius = 0; area = 0; cmp = 0.f;
    x0 = x1 = y0 = y1 = 0;
  }
  xys[i] = xy;
  depths[i] = depth;
  radii[i] = radius;
  conics[3 * i] = conic.x; conics[3 * i + 1] = conic.y; conics[3 * i + 2] = conic.z;
  comp_out[i] = cmp;
  tiles_hit[i] = area;
  tile_box[4 * i] = x0; tile_box[4 * i + 1] = y0; tile_box[4 * i + 2] = x1; tile_box[4 * i + 3] = y1;
  if (!ok) {
    tbox[4 * i] = tbox[4 * i + 1] = tbox[4 * i + 2] = tbox[4 * i + 3] = 0;
    thits[i] = 0;
    return;
  }
  // view-dependent colour (splatfacto.py:769-777): clamp(SH + 0.5, min 0); degree < 0 means "no SH": sigmoid of the DC term
  float dx = mx - cam.pos[0], dy = my - cam.pos[1], dz = mz - cam.pos[2];
  float dn = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  dx *= dn; dy *= dn; dz *= dn;
  float col[4];
  const float* rest = s_rest + (int)threadIdx.x * rest_coeffs * 3;
  const float* trest = s_trest + (int)threadIdx.x * rest_coeffs;
  if (sh_degree >= 0) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) col[ch] = fmaxf(sh_eval(sh_degree, dx, dy, dz, f_dc + 3 * i, rest, 3, ch) + 0.5f, 0.0f);
    col[3] = fmaxf(sh_eval(sh_degree, dx, dy, dz, t_dc + i, trest, 1, 0) + 0.5f, 0.0f);
  } else {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) col[ch] = 1.0f / (1.0f + expf(-f_dc[3 * i + ch]));
    col[3] = 1.0f / (1.0f + expf(-t_dc[i]));
  }
  float op = 1.0f / (1.0f + expf(-opac_logit[i]));
  // Where can this Gaussian reach alpha >= 1/255 at all?  alpha = opacity exp(-sigma) >= 1/255  <=>  sigma <= ln(255 opacity) =: s_max, and
  // {sigma <= s_max} is the ellipse d^T cov2d^-1 d <= 2 s_max, whose bounding box has half extents sqrt(2 s_max cov_xx), sqrt(2 s_max cov_yy).
  // Tiles (and, in the rasteriser, 8x8 pixel quadrants) outside that box see nothing of the Gaussian: dropping them is EXACT.  The box is
  // intersected with gsplat's 3-sigma tile box (pixels outside THAT never see the Gaussian in the reference even where alpha >= 1/255).
  // In antialiased mode the depth pass uses the plain opacity (>= the compensated one): the bound uses the larger.
  float smax = logf(255.0f * op);
  float hx = -1.0f, hy = -1.0f;
  int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
  if (smax > 0.0f) {
    float inv_cd = 1.0f / (conic.x * conic.z - conic.y * conic.y);  // cov2d = conic^-1: cov_xx = conic.z / det, cov_yy = conic.x / det
    hx = sqrtf(2.0f * smax * conic.z * inv_cd) * 1.0005f + 1e-3f;
    hy = sqrtf(2.0f * smax * conic.x * inv_cd) * 1.0005f + 1e-3f;
    // tile t holds the pixel centres 16 t + 0.5 .. 16 t + 15.5
    bx0 = max(x0, (int)ceilf((xy.x - hx - 15.5f) / (float)SPLAT_BLOCK));
    bx1 = min(x1, (int)floorf((xy.x + hx - 0.5f) / (float)SPLAT_BLOCK) + 1);
    by0 = max(y0, (int)ceilf((xy.y - hy - 15.5f) / (float)SPLAT_BLOCK));
    by1 = min(y1, (int)floorf((xy.y + hy - 0.5f) / (float)SPLAT_BLOCK) + 1);
    if (bx1 <= bx0 || by1 <= by0) bx0 = bx1 = by0 = by1 = 0;
  }
  tbox[4 * i] = bx0; tbox[4 * i + 1] = by0; tbox[4 * i + 2] = bx1; tbox[4 * i + 3] = by1;
  thits[i] = (bx1 - bx0) * (by1 - by0);
  const float LOG2E = 1.4426950408889634f;
  SplatRec r;
  r.a = make_float4(xy.x, xy.y, 0.5f * conic.x * LOG2E, conic.y * LOG2E);
  r.b = make_float4(0.5f * conic.z * LOG2E, log2f(antialiased ? op * cmp : op), hx, hy);
  r.c = make_float4(col[0], col[1], col[2], col[3]);
  r.d = make_float4(depth, log2f(op), 0.f, 0.f);
  recs[i] = r;
}

// ------------------------------------------------------------------------------------------------ tile binning
// One wave = 64 consecutive Gaussians of the depth order.  Their pairs occupy ONE contiguous range of the output (cum is the inclusive scan
// in the same order), so the lanes walk that range position by position -- coalesced 4-byte stores -- and find the owning Gaussian of a
// position by binary search over the wave's 64 range ends in LDS (a thread-per-Gaussian loop writes 64 scattered streams and serialises
// on the Gaussians that cover hundreds of tiles).
__global__ void __launch_bounds__(256) k_splat_intersect(const int32_t* __restrict__ order, const int32_t* __restrict__ tile_box,
                                                         const int32_t* __restrict__ cum, int64_t N, int tbx, uint32_t* __restrict__ keys,
                                                         int32_t* __restrict__ vals, int64_t capacity) {
  __shared__ int32_t s_end[4][64], s_x0[4][64], s_y0[4][64], s_w[4][64], s_id[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t j0 = (blockIdx.x * (int64_t)(blockDim.x >> 6) + wv) * 64;
  if (j0 >= N) return;  // whole wave
  const int64_t j = j0 + lane;
  int32_t end = 0, g = 0, x0 = 0, y0 = 0, w = 1;
  if (j < N) {
    end = cum[j];
    g = order[j];
    x0 = tile_box[4 * g]; y0 = tile_box[4 * g + 1];
    w = max(tile_box[4 * g + 2] - x0, 1);
  } else {
    end = cum[N - 1];
  }
  s_end[wv][lane] = end; s_x0[wv][lane] = x0; s_y0[wv][lane] = y0; s_w[wv][lane] = w; s_id[wv][lane] = g;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are visible to its other lanes
  const int64_t begin = j0 == 0 ? 0 : cum[j0 - 1];
  const int64_t stop = s_end[wv][63];
  for (int64_t pos = begin + lane; pos < stop; pos += 64) {
    // first lane o with end_o > pos
    int lo = 0, hi = 63;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if ((int64_t)s_end[wv][mid] > pos) hi = mid; else lo = mid + 1;
    }
    int64_t first = lo == 0 ? begin : (int64_t)s_end[wv][lo - 1];
    int local = (int)(pos - first);
    int ww = s_w[wv][lo];
    int ty = local / ww, tx = local - ty * ww;
    if (pos < capacity) {
      keys[pos] = (uint32_t)((s_y0[wv][lo] + ty) * tbx + s_x0[wv][lo] + tx);
      vals[pos] = s_id[wv][lo];
    }
  }
}

__global__ void k_splat_tile_edges(const uint32_t* __restrict__ keys, int64_t M, int32_t* __restrict__ tile_bins) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= M) return;
  int32_t t = (int32_t)keys[i];
  if (i == 0) tile_bins[2 * t] = 0;
  else {
    int32_t tp = (int32_t)keys[i - 1];
    if (tp != t) { tile_bins[2 * tp + 1] = (int32_t)i; tile_bins[2 * t] = (int32_t)i; }
  }
  if (i == M - 1) tile_bins[2 * t + 1] = (int32_t)M;
}

// longest tiles first: a tile is a sequential front-to-back walk, so the frame ends when the deepest tile ends -- started last (row-major
// order puts the deep centre of the image in the middle of the launch) it runs alone at the end with the rest of the chip idle
__global__ void k_splat_tile_len(const int32_t* __restrict__ tile_bins, int num_tiles, uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= num_tiles) return;
  keys[t] = 0xffffffffu - (uint32_t)(tile_bins[2 * t + 1] - tile_bins[2 * t]);
  vals[t] = t;
}

// ------------------------------------------------------------------------------------------------ rasteriser
template <bool AA>
__global__ void __launch_bounds__(256) k_splat_raster(const SplatRec* __restrict__ recs, const int32_t* __restrict__ sorted_ids,
                                                      const int32_t* __restrict__ tile_bins, const int32_t* __restrict__ tile_order, int W, int H,
                                                      int tbx, float4 background, float* __restrict__ out_rgbt, float* __restrict__ out_depth,
                                                      float* __restrict__ out_alpha, uint32_t* __restrict__ depth_max) {
  __shared__ float4 sa[SPLAT_BATCH], sb[SPLAT_BATCH], sc[SPLAT_BATCH], sd[SPLAT_BATCH];
  __shared__ float smax[4];
  const int tile = tile_order[blockIdx.x];
  const int tile_x = tile % tbx, tile_y = tile / tbx;
  // lane -> pixel: a wave covers one 8x8 QUADRANT of the tile (the squarest 64-pixel footprint: the per-wave culling below rejects the most
  // Gaussians for it); 8 lanes = one 128-byte row segment of the RGBT output
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qx0 = tile_x * SPLAT_BLOCK + 8 * (wv & 1), qy0 = tile_y * SPLAT_BLOCK + 8 * (wv >> 1);
  const int ix = qx0 + (lane & 7), iy = qy0 + (lane >> 3);
  const bool inside = ix < W && iy < H;
  const float pxf = (float)ix + 0.5f, pyf = (float)iy + 0.5f;
  const float qcx = (float)qx0 + 4.0f, qcy = (float)qy0 + 4.0f;  // centre of the quadrant's pixel centres (they span +-3.5 around it)
  const int begin = tile_bins[2 * tile], end = tile_bins[2 * tile + 1];
  float T = 1.0f, Td = 1.0f;
  f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
  float dacc = 0.f;
  bool done = !inside, done_d = !inside;
  // software pipeline: the records of batch i+1 are fetched (two dependent gathers: id, then the 64-byte record) while batch i is blended
  float4 ra, rb, rc, rd;
  ra = rb = rc = rd = make_float4(0.f, 0.f, 0.f, 0.f);
  if (begin + (int)threadIdx.x < end) {
    const SplatRec* r = recs + sorted_ids[begin + (int)threadIdx.x];
    ra = r->a; rb = r->b; rc = r->c; rd = r->d;
  }
  for (int base = begin; base < end; base += SPLAT_BATCH) {
    // all 256 pixels finished -> nothing left to blend in this tile
    if (__syncthreads_count((AA ? (done && done_d) : done) ? 1 : 0) == 256) break;
    sa[threadIdx.x] = ra; sb[threadIdx.x] = rb; sc[threadIdx.x] = rc; sd[threadIdx.x] = rd;
    __syncthreads();
    {
      int nidx = base + SPLAT_BATCH + (int)threadIdx.x;
      if (nidx < end) {
        const SplatRec* r = recs + sorted_ids[nidx];
        ra = r->a; rb = r->b; rc = r->c; rd = r->d;
      }
    }
    const int n = min(SPLAT_BATCH, end - base);
    // the whole wave is done: skip the arithmetic of this batch (wave-uniform branch)
    if (__all((AA ? (done && done_d) : done) ? 1 : 0)) continue;
    // Per-wave culling, 64 Gaussians per step: lane l tests Gaussian 64 q + l against this wave's quadrant (box around {alpha >= 1/255},
    // see k_splat_project); the ballot is the list of Gaussians that can touch the quadrant, walked with scalar bit operations.  A
    // Gaussian that misses the quadrant costs 1/64 of a test instead of a full evaluation on all 64 lanes.
#pragma unroll 1
    for (int q = 0; q < SPLAT_BATCH / 64; ++q) {
      if (q * 64 >= n) break;
      const int kk = q * 64 + lane;
      bool keep = false;
      if (kk < n) {
        float4 a = sa[kk], b = sb[kk];
        keep = fabsf(a.x - qcx) <= b.z + 3.5f && fabsf(a.y - qcy) <= b.w + 3.5f;
      }
      uint64_t live = __ballot(keep);
      while (live) {
        const int k = q * 64 + __builtin_ctzll(live);
        live &= live - 1;
        // wave-uniform (broadcast) LDS reads.  Holding the records in registers and broadcasting the fields with v_readlane instead was
        // measured slower (403 vs 360 us: +11 VALU instructions per Gaussian in a kernel that is VALU-issue bound).
        float4 a = sa[k], b = sb[k];
        float dx = a.x - pxf, dy = a.y - pyf;
        float power = fmaf(dx, fmaf(a.z, dx, a.w * dy), b.x * dy * dy);  // sigma log2(e)
        float ex = b.y - power;
        // nobody in the wave can reach alpha >= 1/255 = 2^-7.994 (or the form is negative): skip the exponential and the blend
        // (antialiased: the depth pass blends with the plain opacity, which is the larger one)
        if (!__any((power >= 0.f && (AA ? sd[k].y - power : ex) >= -8.0f) ? 1 : 0)) continue;
        if (power < 0.f) continue;
        float alpha = fminf(0.999f, __builtin_amdgcn_exp2f(ex));
        if (!done && alpha >= (1.0f / 255.0f)) {
          float nT = fmaf(-alpha, T, T);
          if (nT <= 1e-4f) done = true;
          else {
            float vis = alpha * T;
            float4 c = sc[k];
            f32x2 v2 = {vis, vis};
            acc01 = __builtin_elementwise_fma(v2, (f32x2){c.x, c.y}, acc01);
            acc23 = __builtin_elementwise_fma(v2, (f32x2){c.z, c.w}, acc23);
            if (!AA) dacc = fmaf(vis, sd[k].x, dacc);
            T = nT;
          }
        }
        if (AA) {
          float4 d = sd[k];
          float alpha_d = fminf(0.999f, __builtin_amdgcn_exp2f(d.y - power));
          if (!done_d && alpha_d >= (1.0f / 255.0f)) {
            float nT = fmaf(-alpha_d, Td, Td);
            if (nT <= 1e-4f) done_d = true;
            else { dacc = fmaf(alpha_d * Td, d.x, dacc); Td = nT; }
          }
        }
      }
    }
  }
  const float acc[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
  float dmax = 0.f;
  if (inside) {
    int64_t p = (int64_t)iy * W + ix;
    float4 o = make_float4(fminf(acc[0] + T * background.x, 1.0f), fminf(acc[1] + T * background.y, 1.0f), fminf(acc[2] + T * background.z, 1.0f),
                           fminf(acc[3] + T * background.w, 1.0f));
    reinterpret_cast<float4*>(out_rgbt)[p] = o;
    out_alpha[p] = 1.0f - T;
    out_depth[p] = dacc;  // un-normalised: k_splat_depth_finalize divides by alpha
    dmax = dacc;
  }
  // max of the un-normalised depth image (the reference fills alpha == 0 pixels with it, splatfacto.py:809); depths are >= 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = dmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    if (m > 0.f) atomicMax(depth_max, __float_as_uint(m));
  }
}

__global__ void k_splat_depth_finalize(float* __restrict__ depth, const float* __restrict__ alpha, const uint32_t* __restrict__ depth_max, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = alpha[i];
  depth[i] = a > 0.f ? depth[i] / a : __uint_as_float(*depth_max);
}

// ------------------------------------------------------------------------------------------------ entry points
static int check_cam(const TnSplatCamera* cam, const char* who) {
  TN_REQUIRE(cam != nullptr, "%s: null camera", who);
  TN_REQUIRE(cam->width >= 1 && cam->height >= 1 && cam->width <= 16384 && cam->height <= 16384, "%s: bad image size %dx%d", who, cam->width, cam->height);
  TN_REQUIRE(cam->fx > 0.f && cam->fy > 0.f, "%s: bad focal length", who);
  return TN_OK;
}
static SplatCamK make_camk(const TnSplatCamera* cam) {
  SplatCamK k;
  for (int i = 0; i < 12; ++i) k.view[i] = cam->viewmat[i];
  for (int i = 0; i < 16; ++i) k.proj[i] = cam->projmat[i];
  k.fx = cam->fx; k.fy = cam->fy; k.cx = cam->cx; k.cy = cam->cy; k.clip = cam->clip_thresh;
  for (int i = 0; i < 3; ++i) k.pos[i] = cam->position[i];
  k.W = cam->width; k.H = cam->height;
  k.tbx = (cam->width + SPLAT_BLOCK - 1) / SPLAT_BLOCK;
  k.tby = (cam->height + SPLAT_BLOCK - 1) / SPLAT_BLOCK;
  return k;
}

extern "C" int tn_splat_project(const TnSplatCamera* camera, const float* means, const float* log_scales, const float* quats, const float* opacities,
                                const float* features_dc, const float* features_rest, const float* thermal_dc, const float* thermal_rest,
                                int64_t num_gaussians, int32_t num_rest_coeffs, int32_t sh_degree, int32_t antialiased, float* xys, float* depths,
                                int32_t* radii, float* conics, float* compensation, int32_t* num_tiles_hit, int32_t* tile_box, void* workspace,
                                int64_t max_intersections, tn_stream_t stream) {
  int rc = check_cam(camera, "tn_splat_project");
  if (rc) return rc;
  if (num_gaussians == 0) return TN_OK;
  TN_REQUIRE(means && log_scales && quats && opacities && features_dc && thermal_dc && xys && depths && radii && conics && compensation &&
                 num_tiles_hit && tile_box && workspace,
             "tn_splat_project: null pointer");
  TN_REQUIRE(num_gaussians > 0 && num_gaussians < (1ll << 31), "tn_splat_project: bad Gaussian count");
  TN_REQUIRE(sh_degree >= -1 && sh_degree <= 3, "tn_splat_project: sh_degree %d unsupported (-1 = sigmoid of the DC term, 0..3)", sh_degree);
  TN_REQUIRE(num_rest_coeffs >= (sh_degree < 1 ? 0 : (sh_degree + 1) * (sh_degree + 1) - 1), "tn_splat_project: %d higher-order coefficients for degree %d",
             num_rest_coeffs, sh_degree);
  TN_REQUIRE(num_rest_coeffs == 0 || (features_rest && thermal_rest), "tn_splat_project: null SH coefficients");
  SplatCamK k = make_camk(camera);
  SplatWs ws = splat_layout(workspace, num_gaussians, max_intersections, k.tbx * k.tby, nullptr);
  const int PB = 128;  // Gaussians per block: 30 KB of LDS at degree 3 -> 5 blocks per CU (256 per block = 61 KB = 2 blocks: 136 vs 1xx us)
  const size_t lds = (size_t)PB * num_rest_coeffs * 4 * sizeof(float);
  TN_REQUIRE(lds <= 65536, "tn_splat_project: %d higher-order coefficients do not fit the LDS staging", num_rest_coeffs);
  hipLaunchKernelGGL(k_splat_project, dim3((unsigned)tn_cdiv(num_gaussians, PB)), dim3(PB), lds, tn_s(stream), k, means, log_scales, quats, opacities,
                     features_dc, features_rest, thermal_dc, thermal_rest, num_gaussians, sh_degree, num_rest_coeffs, antialiased, (float2*)xys, depths, radii,
                     conics, compensation, num_tiles_hit, tile_box, ws.recs, ws.tbox, ws.thits);
  TN_CHECK_LAUNCH("tn_splat_project");
  return TN_OK;
}

extern "C" int tn_splat_bin(const TnSplatCamera* camera, const float* depths, int64_t num_gaussians, void* workspace, int64_t max_intersections,
                            int64_t* num_intersections_out, tn_stream_t stream) {
  int rc = check_cam(camera, "tn_splat_bin");
  if (rc) return rc;
  TN_REQUIRE(num_intersections_out != nullptr, "tn_splat_bin: null output");
  *num_intersections_out = 0;
  SplatCamK k = make_camk(camera);
  const int num_tiles = k.tbx * k.tby;
  TN_REQUIRE(workspace != nullptr, "tn_splat_bin: null workspace");
  SplatWs ws = splat_layout(workspace, num_gaussians, max_intersections, num_tiles, nullptr);
  hipStream_t st = tn_s(stream);
  if (hipMemsetAsync(ws.tile_bins, 0, 8 * (size_t)num_tiles, st) != hipSuccess || hipMemsetAsync(ws.depth_max, 0, 4, st) != hipSuccess) {
    tn_set_error("tn_splat_bin: memset failed");
    return TN_ELAUNCH;
  }
  // default tile order = identity (every run empty); replaced by longest-first once the runs are known
  hipLaunchKernelGGL(k_splat_tile_len, dim3((unsigned)tn_cdiv(num_tiles, 256)), dim3(256), 0, st, ws.tile_bins, num_tiles, ws.lkeys[1], ws.lvals[1]);
  TN_CHECK_LAUNCH("tn_splat_bin(order)");
  if (num_gaussians == 0) return TN_OK;
  TN_REQUIRE(depths != nullptr, "tn_splat_bin: null pointer");
  // depth order of the Gaussians (stable: equal depths keep their index order; culled Gaussians have depth 0 and no tiles)
  size_t tb = ws.tmp_bytes;
  if (rocprim::radix_sort_pairs(ws.tmp, tb, reinterpret_cast<const uint32_t*>(depths), ws.dkeys, rocprim::counting_iterator<int32_t>(0), ws.order,
                                (size_t)num_gaussians, 0, 32, st) != hipSuccess) {
    tn_set_error("tn_splat_bin: depth sort failed");
    return TN_ELAUNCH;
  }
  tb = ws.tmp_bytes;
  if (rocprim::inclusive_scan(ws.tmp, tb, rocprim::make_transform_iterator((const int32_t*)ws.order, HitsInOrder{ws.thits}), ws.cum,
                              (size_t)num_gaussians, rocprim::plus<int32_t>(), st) != hipSuccess) {
    tn_set_error("tn_splat_bin: scan failed");
    return TN_ELAUNCH;
  }
  // the number of (Gaussian, tile) pairs sizes the sort: one 4-byte read-back, as gsplat's own binning does (cum_tiles_hit[-1].item())
  int32_t total = 0;
  if (hipMemcpyAsync(&total, ws.cum + (num_gaussians - 1), 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    tn_set_error("tn_splat_bin: read-back of the intersection count failed");
    return TN_ELAUNCH;
  }
  *num_intersections_out = total;
  TN_REQUIRE(total >= 0, "tn_splat_bin: intersection count overflowed 2^31");
  if (total > max_intersections) {
    tn_set_error("tn_splat_bin: %d intersections exceed the workspace capacity %lld (call again with a larger workspace)", total, (long long)max_intersections);
    return TN_EINVAL;
  }
  if (total == 0) return TN_OK;
  hipLaunchKernelGGL(k_splat_intersect, dim3((unsigned)tn_cdiv(num_gaussians, 256)), dim3(256), 0, st, ws.order, ws.tbox, ws.cum, num_gaussians, k.tbx,
                     ws.keys[0], ws.vals[0], max_intersections);
  TN_CHECK_LAUNCH("tn_splat_bin(intersect)");
  int tile_bits = 1;
  while ((1 << tile_bits) < num_tiles) ++tile_bits;
  tb = ws.tmp_bytes;
  if (rocprim::radix_sort_pairs(ws.tmp, tb, ws.keys[0], ws.keys[1], ws.vals[0], ws.vals[1], (size_t)total, 0, tile_bits, st) != hipSuccess) {
    tn_set_error("tn_splat_bin: radix sort failed");
    return TN_ELAUNCH;
  }
  hipLaunchKernelGGL(k_splat_tile_edges, dim3((unsigned)tn_cdiv(total, 256)), dim3(256), 0, st, ws.keys[1], (int64_t)total, ws.tile_bins);
  TN_CHECK_LAUNCH("tn_splat_bin(edges)");
  hipLaunchKernelGGL(k_splat_tile_len, dim3((unsigned)tn_cdiv(num_tiles, 256)), dim3(256), 0, st, ws.tile_bins, num_tiles, ws.lkeys[0], ws.lvals[0]);
  TN_CHECK_LAUNCH("tn_splat_bin(lengths)");
  tb = ws.tmp_bytes;
  if (rocprim::radix_sort_pairs(ws.tmp, tb, ws.lkeys[0], ws.lkeys[1], ws.lvals[0], ws.lvals[1], (size_t)num_tiles, 0, 32, st) != hipSuccess) {
    tn_set_error("tn_splat_bin: tile order sort failed");
    return TN_ELAUNCH;
  }
  return TN_OK;
}

extern "C" int tn_splat_raster(const TnSplatCamera* camera, int64_t num_gaussians, void* workspace, int64_t max_intersections, const float* background4,
                               int32_t antialiased, float* out_rgbt, float* out_depth, float* out_alpha, tn_stream_t stream) {
  int rc = check_cam(camera, "tn_splat_raster");
  if (rc) return rc;
  TN_REQUIRE(workspace && background4 && out_rgbt && out_depth && out_alpha, "tn_splat_raster: null pointer");
  SplatCamK k = make_camk(camera);
  SplatWs ws = splat_layout(workspace, num_gaussians, max_intersections, k.tbx * k.tby, nullptr);
  float4 bg = make_float4(background4[0], background4[1], background4[2], background4[3]);
  hipStream_t st = tn_s(stream);
  if (antialiased)
    hipLaunchKernelGGL(k_splat_raster<true>, dim3(k.tbx * k.tby), dim3(256), 0, st, ws.recs, ws.vals[1], ws.tile_bins, ws.lvals[1], k.W, k.H, k.tbx, bg, out_rgbt, out_depth,
                       out_alpha, ws.depth_max);
  else
    hipLaunchKernelGGL(k_splat_raster<false>, dim3(k.tbx * k.tby), dim3(256), 0, st, ws.recs, ws.vals[1], ws.tile_bins, ws.lvals[1], k.W, k.H, k.tbx, bg, out_rgbt, out_depth,
                       out_alpha, ws.depth_max);
  TN_CHECK_LAUNCH("tn_splat_raster");
  int64_t n = (int64_t)k.W * k.H;
  hipLaunchKernelGGL(k_splat_depth_finalize, dim3((unsigned)tn_cdiv(n, 256)), dim3(256), 0, st, out_depth, out_alpha, ws.depth_max, n);
  TN_CHECK_LAUNCH("tn_splat_raster(depth)");
  return TN_OK;
}
