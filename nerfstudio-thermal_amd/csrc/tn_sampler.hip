// Per-ray kernels: one 64-lane wave owns one ray; a level's <=256 samples sit ITEMS-per-lane in registers and
// the serial dependences of the reference (transmittance cumsum, CDF cumsum, searchsorted) become wave-wide scans.
//   tn_spaced_bins   - SpacedSampler / UniformLinDispPiecewiseSampler   (model_components/ray_samplers.py:78-128,225-248)
//   tn_weights_fwd/bwd - RaySamples.get_weights (+ median depth)        (cameras/rays.py:128-150; renderers.py:547-557)
//   tn_pdf_resample  - PDFSampler (+ anneal pow)                         (model_components/ray_samplers.py:276-372,602)
//   tn_composite_*   - RGB(T)Renderer / Accumulation / Depth renderers   (model_components/renderers.py:118-133,238-245,509,547-576)
//   tn_distortion_loss / tn_interlevel_loss                              (model_components/losses.py:57-158)
#include "tn_common.h"
#include "tn_pixel_loss.h"
#include "tn_sampler_ray.h"

#define RAYS_PER_BLOCK 4
#define BLOCK (RAYS_PER_BLOCK * TN_WAVE)

// ------------------------------------------------------------------------------------------------ spaced bins
__global__ void k_spaced_bins(const float* __restrict__ lin_bins, const float* __restrict__ jitter, const float* __restrict__ nears,
                              const float* __restrict__ fars, int64_t N, int S, float* __restrict__ s_bins, float* __restrict__ e_bins) {
  tn_spaced_bins_body(lin_bins, jitter, nears, fars, N, S, s_bins, e_bins, blockIdx.x, gridDim.x);
}

extern "C" int tn_spaced_bins(const float* lin_bins, const float* jitter, const float* nears, const float* fars, int64_t N, int32_t S,
                              float* s_bins, float* e_bins, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(lin_bins && nears && fars && s_bins && e_bins, "tn_spaced_bins: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_spaced_bins: bad N=%lld S=%d", (long long)N, S);
  if (N == 0) return TN_OK;
  int grid = (int)std::min<int64_t>(tn_cdiv(N, 4), 4096);  // one wave per ray, 4 rays per block
  hipLaunchKernelGGL(k_spaced_bins, dim3(grid), dim3(256), 0, tn_s(stream), lin_bins, jitter, nears, fars, N, S, s_bins, e_bins);
  TN_CHECK_LAUNCH("tn_spaced_bins");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ weights
template <int ITEMS, int C>
__device__ __forceinline__ void ray_load_rgb(const float* __restrict__ rgb, int S, int64_t ray, int lane, float (&v)[ITEMS][C], float (&last)[C]) {
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
#pragma unroll
    for (int c = 0; c < C; ++c) v[k][c] = (i < S) ? rgb[(ray * S + i) * C + c] : 0.0f;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) last[c] = rgb[(ray * S + (S - 1)) * C + c];
}
template <int ITEMS>
__global__ void __launch_bounds__(BLOCK) k_weights_fwd(const float* __restrict__ e_bins, const float* __restrict__ density, int64_t N, int S,
                                                       float* __restrict__ weights, float* __restrict__ median_depth) {
  int lane = tn_lane();
  int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
  if (ray >= N) return;  // whole wave exits together
  float w[ITEMS];
  weights_body<ITEMS>(e_bins, density, S, ray, weights, median_depth, lane, w);
}

extern "C" int tn_weights_fwd(const float* e_bins, const float* density, int64_t N, int32_t S, float* weights, float* median_depth,
                              tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(e_bins && density && weights, "tn_weights_fwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_weights_fwd: bad N=%lld S=%d", (long long)N, S);
  if (N == 0) return TN_OK;
  dim3 grid((unsigned)tn_cdiv(N, RAYS_PER_BLOCK)), block(BLOCK);
  if (S <= 64) hipLaunchKernelGGL(k_weights_fwd<1>, grid, block, 0, tn_s(stream), e_bins, density, N, S, weights, median_depth);
  else if (S <= 128) hipLaunchKernelGGL(k_weights_fwd<2>, grid, block, 0, tn_s(stream), e_bins, density, N, S, weights, median_depth);
  else hipLaunchKernelGGL(k_weights_fwd<4>, grid, block, 0, tn_s(stream), e_bins, density, N, S, weights, median_depth);
  TN_CHECK_LAUNCH("tn_weights_fwd");
  return TN_OK;
}

// backward: w_i = (1-exp(-dd_i)) * exp(-sum_{k<i} dd_k)
//   d dd_i = dw_i * exp(-dd_i) * T_i  -  sum_{k>i} dw_k * w_k ;   d density_i = d dd_i * delta_i
// (nan_to_num passes the gradient where the raw weight is finite, which is every finite-input case.)
template <int ITEMS>
__global__ void __launch_bounds__(BLOCK) k_weights_bwd(const float* __restrict__ e_bins, const float* __restrict__ density,
                                                       const float* __restrict__ weights, const float* __restrict__ d_weights, int64_t N, int S,
                                                       float* __restrict__ d_density) {
  int lane = tn_lane();
  int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
  if (ray >= N) return;
  const float* eb = e_bins + ray * (S + 1);
  float dd[ITEMS], delta[ITEMS], gw[ITEMS], wk[ITEMS];
  double loc = 0.0;
  float sloc = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    dd[k] = delta[k] = gw[k] = wk[k] = 0.0f;
    if (i < S) {
      delta[k] = eb[i + 1] - eb[i];
      dd[k] = delta[k] * density[ray * S + i];
      gw[k] = d_weights[ray * S + i];
      wk[k] = weights[ray * S + i];
    }
    loc += (double)dd[k];
    sloc += gw[k] * wk[k];
  }
  double incl = tn_wave_incl_scan_d(loc, lane);
  double run = tn_excl_from_incl_d(incl, lane);
  float sincl = tn_wave_incl_rscan(sloc, lane);
  float suffix = tn_rexcl_from_incl(sincl, lane);  // sum over later lanes
  // inside the lane: walk backwards for the suffix part
  float T[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) { T[k] = expf(-(float)run); run += (double)dd[k]; }
#pragma unroll
  for (int k = ITEMS - 1; k >= 0; --k) {
    int i = lane * ITEMS + k;
    float g = gw[k] * expf(-dd[k]) * T[k] - suffix;
    if (i < S) d_density[ray * S + i] = g * delta[k];
    suffix += gw[k] * wk[k];
  }
}

extern "C" int tn_weights_bwd(const float* e_bins, const float* density, const float* weights, const float* d_weights, int64_t N, int32_t S,
                              float* d_density, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(e_bins && density && weights && d_weights && d_density, "tn_weights_bwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_weights_bwd: bad N=%lld S=%d", (long long)N, S);
  if (N == 0) return TN_OK;
  dim3 grid((unsigned)tn_cdiv(N, RAYS_PER_BLOCK)), block(BLOCK);
  if (S <= 64) hipLaunchKernelGGL(k_weights_bwd<1>, grid, block, 0, tn_s(stream), e_bins, density, weights, d_weights, N, S, d_density);
  else if (S <= 128) hipLaunchKernelGGL(k_weights_bwd<2>, grid, block, 0, tn_s(stream), e_bins, density, weights, d_weights, N, S, d_density);
  else hipLaunchKernelGGL(k_weights_bwd<4>, grid, block, 0, tn_s(stream), e_bins, density, weights, d_weights, N, S, d_density);
  TN_CHECK_LAUNCH("tn_weights_bwd");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ PDF resample
template <int ITEMS>
__global__ void __launch_bounds__(BLOCK) k_pdf_resample(const float* __restrict__ s_bins_prev, const float* __restrict__ weights_prev, int Sp,
                                                        float anneal, const float* __restrict__ u_lin, const float* __restrict__ jitter,
                                                        const float* __restrict__ nears, const float* __restrict__ fars, int64_t N, int S,
                                                        float* __restrict__ s_bins, float* __restrict__ e_bins) {
  __shared__ float sh_cdf[RAYS_PER_BLOCK][TN_MAX_SAMPLES + 1];
  __shared__ float sh_bins[RAYS_PER_BLOCK][TN_MAX_SAMPLES + 1];
  int lane = tn_lane();
  int wv = threadIdx.x >> 6;
  int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
  if (ray >= N) return;  // no block-level barrier below: waves are independent
  float w_raw[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    w_raw[k] = (i < Sp) ? weights_prev[ray * Sp + i] : 0.0f;
  }
  PdfLoads L;
  pdf_load(s_bins_prev, Sp, u_lin, jitter, nears, fars, S, ray, lane, L);
  pdf_body<ITEMS>(w_raw, L, Sp, anneal, jitter != nullptr, S, ray, s_bins, e_bins, sh_cdf[wv], sh_bins[wv], lane);
}
// RaySamples.get_weights of a proposal level and the PDF resampling it feeds, one wave per ray, the weights handed over in registers
// (the two are always called back to back by ProposalNetworkSampler.generate_ray_samples, ray_samplers.py:593-611).
template <int ITEMS>
__global__ void __launch_bounds__(BLOCK) k_weights_pdf(const float* __restrict__ e_bins_prev, const float* __restrict__ density_prev,
                                                       const float* __restrict__ s_bins_prev, int Sp, float anneal,
                                                       const float* __restrict__ u_lin, const float* __restrict__ jitter,
                                                       const float* __restrict__ nears, const float* __restrict__ fars, int64_t N, int S,
                                                       float* __restrict__ weights_prev, float* __restrict__ median_prev,
                                                       float* __restrict__ s_bins, float* __restrict__ e_bins) {
  __shared__ float sh_cdf[RAYS_PER_BLOCK][TN_MAX_SAMPLES + 1];
  __shared__ float sh_bins[RAYS_PER_BLOCK][TN_MAX_SAMPLES + 1];
  int lane = tn_lane();
  int wv = threadIdx.x >> 6;
  int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
  if (ray >= N) return;
  float w[ITEMS];
  PdfLoads L;
  pdf_load(s_bins_prev, Sp, u_lin, jitter, nears, fars, S, ray, lane, L);  // in flight beside the loads of get_weights
  weights_body<ITEMS>(e_bins_prev, density_prev, Sp, ray, weights_prev, median_prev, lane, w);
  pdf_body<ITEMS>(w, L, Sp, anneal, jitter != nullptr, S, ray, s_bins, e_bins, sh_cdf[wv], sh_bins[wv], lane);
}

extern "C" int tn_pdf_resample(const float* s_bins_prev, const float* weights_prev, int32_t S_prev, float anneal, const float* u_lin,
                               const float* jitter, const float* nears, const float* fars, int64_t N, int32_t S, float* s_bins,
                               float* e_bins, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(s_bins_prev && weights_prev && u_lin && nears && fars && s_bins && e_bins, "tn_pdf_resample: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES && S_prev >= 1 && S_prev <= TN_MAX_SAMPLES, "tn_pdf_resample: bad N=%lld S=%d S_prev=%d",
             (long long)N, S, S_prev);
  if (N == 0) return TN_OK;
  dim3 grid((unsigned)tn_cdiv(N, RAYS_PER_BLOCK)), block(BLOCK);
#define LAUNCH_PDF(I)                                                                                                                \
  hipLaunchKernelGGL(k_pdf_resample<I>, grid, block, 0, tn_s(stream), s_bins_prev, weights_prev, S_prev, anneal, u_lin, jitter, nears, \
                     fars, N, S, s_bins, e_bins)
  if (S_prev <= 64) LAUNCH_PDF(1);
  else if (S_prev <= 128) LAUNCH_PDF(2);
  else LAUNCH_PDF(4);
#undef LAUNCH_PDF
  TN_CHECK_LAUNCH("tn_pdf_resample");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ composite
__device__ __forceinline__ uint32_t f2ord(float f) { return __float_as_uint(f); }  // midpoints are >= 0: bit pattern is order-preserving

__global__ void k_minmax_init(uint32_t* mm) {
  mm[0] = 0x7f800000u;  // +inf
  mm[1] = 0u;           // 0
}
extern "C" int tn_minmax_init(uint32_t* steps_minmax, tn_stream_t stream) {
  TN_REQUIRE(steps_minmax, "tn_minmax_init: null pointer");
  hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, tn_s(stream), steps_minmax);
  TN_CHECK_LAUNCH("tn_minmax_init");
  return TN_OK;
}

// one ray of the renderers: w[k] = weight of sample lane*ITEMS + k (0 beyond S); mn / mx take the ray's smallest / largest midpoint
template <int ITEMS, int C>
__device__ __forceinline__ void composite_compute(const float (&w)[ITEMS], const float (&st)[ITEMS], const float (&en)[ITEMS], const float (&rv)[ITEMS][C],
                                                  const float (&lastc)[C], int S, int64_t ray, int training, float* __restrict__ comp,
                                                  float* __restrict__ accumulation, float* __restrict__ depth_median, float* __restrict__ depth_expected,
                                                  int lane, float& mn, float& mx, float* comp_out = nullptr) {  // comp_out: [C], valid on lane 0
  float acc_c[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc_c[c] = 0.0f;
  float wsum = 0.0f, wmid = 0.0f;
  mn = INFINITY; mx = 0.0f;
  float mid[ITEMS];
  double wloc = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    mid[k] = 0.0f;
    if (i < S) {
      mid[k] = (st[k] + en[k]) / 2.0f;
      mn = fminf(mn, mid[k]); mx = fmaxf(mx, mid[k]);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float v = rv[k][c];
        if (!training) v = tn_nan_to_num(v);
        acc_c[c] += w[k] * v;
      }
      wsum += w[k];
      wmid += w[k] * mid[k];
    }
    wloc += (double)w[k];
  }
  wsum = tn_wave_sum(wsum);
  wmid = tn_wave_sum(wmid);
#pragma unroll
  for (int c = 0; c < C; ++c) acc_c[c] = tn_wave_sum(acc_c[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float last = lastc[c];
      if (!training) last = tn_nan_to_num(last);
      float v = acc_c[c] + last * (1.0f - wsum);
      if (!training) v = fminf(fmaxf(v, 0.0f), 1.0f);
      comp[ray * C + c] = v;
      if (comp_out != nullptr) comp_out[c] = v;
    }
    if (accumulation) accumulation[ray] = wsum;
    if (depth_expected) depth_expected[ray] = wmid / (wsum + 1e-10f);
  }
  if (depth_median != nullptr) {
    double wincl = tn_wave_incl_scan_d(wloc, lane);
    double wrun = tn_excl_from_incl_d(wincl, lane);
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i = lane * ITEMS + k;
      wrun += (double)w[k];
      if (i < S && (float)wrun < 0.5f) cnt++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    int idx = cnt < S - 1 ? cnt : S - 1;
    int owner = idx / ITEMS, slot = idx - owner * ITEMS;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) if (k == slot) v = mid[k];
    v = __shfl(v, owner, 64);
    if (lane == 0) depth_median[ray] = v;
  }
}

template <int ITEMS, int C>
__device__ __forceinline__ void composite_ray(const float (&w)[ITEMS], const float* __restrict__ rgb, const float* __restrict__ e_bins, int S,
                                              int64_t ray, int training, float* __restrict__ comp, float* __restrict__ accumulation,
                                              float* __restrict__ depth_median, float* __restrict__ depth_expected, int lane, float& mn, float& mx) {
  float st[ITEMS], en[ITEMS], dn[ITEMS], rv[ITEMS][C], lastc[C];
  ray_load_bins<ITEMS>(e_bins, nullptr, S, ray, lane, st, en, dn);
  ray_load_rgb<ITEMS, C>(rgb, S, ray, lane, rv, lastc);
  composite_compute<ITEMS, C>(w, st, en, rv, lastc, S, ray, training, comp, accumulation, depth_median, depth_expected, lane, mn, mx);
}

template <int ITEMS, int C>
__global__ void __launch_bounds__(BLOCK) k_composite_fwd(const float* __restrict__ rgb, const float* __restrict__ weights,
                                                         const float* __restrict__ e_bins, int64_t N, int S, int training,
                                                         float* __restrict__ comp, float* __restrict__ accumulation,
                                                         float* __restrict__ depth_median, float* __restrict__ depth_expected,
                                                         uint32_t* __restrict__ steps_minmax) {
  int lane = tn_lane();
  __shared__ float sh_mn[RAYS_PER_BLOCK], sh_mx[RAYS_PER_BLOCK];
  float blk_mn = INFINITY, blk_mx = 0.0f;
  // grid-stride over groups of RAYS_PER_BLOCK rays: same-address atomics serialise at ~15-25 ns each, so the batch-global min/max (and the
  // loss sums of the kernels below) leave the block once, not once per ray
  for (int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6); ray < N; ray += (int64_t)gridDim.x * RAYS_PER_BLOCK) {
    float w[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i = lane * ITEMS + k;
      w[k] = (i < S) ? weights[ray * S + i] : 0.0f;
    }
    float mn, mx;
    composite_ray<ITEMS, C>(w, rgb, e_bins, S, ray, training, comp, accumulation, depth_median, depth_expected, lane, mn, mx);
    if (steps_minmax != nullptr) {
      blk_mn = fminf(blk_mn, tn_wave_min(mn));
      blk_mx = fmaxf(blk_mx, tn_wave_max(mx));
    }
  }  // ray loop
  if (steps_minmax != nullptr) {
    if (lane == 0) { sh_mn[threadIdx.x >> 6] = blk_mn; sh_mx[threadIdx.x >> 6] = blk_mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = sh_mn[0], b = sh_mx[0];
      for (int w = 1; w < RAYS_PER_BLOCK; ++w) { a = fminf(a, sh_mn[w]); b = fmaxf(b, sh_mx[w]); }
      if (a != INFINITY) { atomicMin(&steps_minmax[0], f2ord(a)); atomicMax(&steps_minmax[1], f2ord(b)); }
    }
  }
}

// ---- get_weights + every renderer of the last level in ONE launch (tn_render_fwd): the weights stay in registers between the two halves.
// The batch-global clip of the expected depth (renderers.py:574: clip to [steps.min(), steps.max()]) needs every block's min / max first:
// block b STORES its pair into scratch[b], scratch[RENDER_MAX_BLOCKS + b] (no atomics: same-address atomics execute one after the other at
// ~25 ns each, a 512-block grid would spend more time on them than on the rendering; and nothing to initialise), and k_clip_depth_blocks
// reduces the pairs and applies the clip.  (A last-block-done epilogue inside this kernel was measured too: the device-scope fences it needs
// write the whole L2 back once per block, 55 us against 20 + 5 us for the two launches.)
#define RENDER_MAX_BLOCKS 512
template <int ITEMS, int C>
__global__ void __launch_bounds__(BLOCK) k_render_fwd(const float* __restrict__ e_bins, const float* __restrict__ density,
                                                      const float* __restrict__ rgb, int64_t N, int S, int training, float* __restrict__ weights,
                                                      float* __restrict__ comp, float* __restrict__ accumulation, float* __restrict__ depth_median,
                                                      float* __restrict__ depth_expected, float* __restrict__ scratch) {
  const int lane = tn_lane();
  __shared__ float sh_mn[RAYS_PER_BLOCK], sh_mx[RAYS_PER_BLOCK];
  float blk_mn = INFINITY, blk_mx = 0.0f;
  for (int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6); ray < N; ray += (int64_t)gridDim.x * RAYS_PER_BLOCK) {
    float w[ITEMS];
    weights_body<ITEMS>(e_bins, density, S, ray, weights, nullptr, lane, w);
    float mn, mx;
    composite_ray<ITEMS, C>(w, rgb, e_bins, S, ray, training, comp, accumulation, depth_median, depth_expected, lane, mn, mx);
    if (depth_expected != nullptr) {
      blk_mn = fminf(blk_mn, tn_wave_min(mn));
      blk_mx = fmaxf(blk_mx, tn_wave_max(mx));
    }
  }
  if (depth_expected == nullptr) return;  // wave-uniform, whole grid
  if (lane == 0) { sh_mn[threadIdx.x >> 6] = blk_mn; sh_mx[threadIdx.x >> 6] = blk_mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = sh_mn[0], b = sh_mx[0];
    for (int w = 1; w < RAYS_PER_BLOCK; ++w) { a = fminf(a, sh_mn[w]); b = fmaxf(b, sh_mx[w]); }
    scratch[blockIdx.x] = a;  // +inf / 0 from a block without rays: neutral
    scratch[RENDER_MAX_BLOCKS + blockIdx.x] = b;
  }
}
// every block reduces the nblk pairs itself (a few KB out of L2), then clips its share of the rays; the maxima start at scratch[hi_off].
// bid / nblocks stand in for blockIdx.x / gridDim.x (the clip also runs as co-work blocks of the loss launch: tn_train_losses_clip).
__device__ __forceinline__ void clip_depth_body(float* __restrict__ d, const float* __restrict__ scratch, int nblk, int hi_off, int64_t N, int bid, int nblocks,
                                                float* sh_lo, float* sh_hi /* 8 floats of LDS each */) {
  const int t = threadIdx.x, nw = blockDim.x >> 6;
  float lo = INFINITY, hi = 0.0f;
  for (int i = t; i < nblk; i += blockDim.x) { lo = fminf(lo, scratch[i]); hi = fmaxf(hi, scratch[hi_off + i]); }
  lo = tn_wave_min(lo); hi = tn_wave_max(hi);
  if ((t & 63) == 0) { sh_lo[t >> 6] = lo; sh_hi[t >> 6] = hi; }
  __syncthreads();
  lo = sh_lo[0]; hi = sh_hi[0];
  for (int w = 1; w < nw; ++w) { lo = fminf(lo, sh_lo[w]); hi = fmaxf(hi, sh_hi[w]); }
  for (int64_t i = bid * (int64_t)blockDim.x + t; i < N; i += (int64_t)nblocks * blockDim.x) {
    float v = d[i];
    if (v == v) v = fminf(fmaxf(v, lo), hi);  // torch.clip; NaN propagates
    d[i] = v;
  }
}
__global__ void __launch_bounds__(512) k_clip_depth_blocks(float* __restrict__ d, const float* __restrict__ scratch, int nblk, int hi_off, int64_t N) {
  __shared__ float sh_lo[8], sh_hi[8];
  clip_depth_body(d, scratch, nblk, hi_off, N, blockIdx.x, gridDim.x, sh_lo, sh_hi);
}

// launch_clip = false: the batch-wide clip of depth_expected is left to the caller (*clip_nblk = the number of min / max pairs in scratch,
// maxima at scratch + RENDER_MAX_BLOCKS): tn_train_step runs it as co-work blocks of the loss launch that follows
int tn_render_fwd_ex(const float* e_bins, const float* density, const float* rgb, int64_t N, int32_t S, int32_t C, int32_t training, float* weights,
                     float* comp, float* accumulation, float* depth_median, float* depth_expected, float* scratch, bool launch_clip, int* clip_nblk,
                     tn_stream_t stream) {
  if (clip_nblk) *clip_nblk = 0;
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(e_bins && density && rgb && weights && comp, "tn_render_fwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_render_fwd: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(C == 1 || C == 3 || C == 4, "tn_render_fwd: unsupported channel count %d", C);
  TN_REQUIRE(depth_expected == nullptr || scratch != nullptr, "tn_render_fwd: depth_expected needs the scratch buffer (TN_RENDER_SCRATCH_FLOATS)");
  static_assert(TN_RENDER_SCRATCH_FLOATS >= 2 * RENDER_MAX_BLOCKS, "render scratch");
  const int nblk = (int)std::min<int64_t>(tn_cdiv(N, RAYS_PER_BLOCK), RENDER_MAX_BLOCKS);
  dim3 grid((unsigned)nblk), block(BLOCK);
  hipStream_t st = tn_s(stream);
#define LAUNCH_R(I, CC)                                                                                                               \
  hipLaunchKernelGGL((k_render_fwd<I, CC>), grid, block, 0, st, e_bins, density, rgb, N, S, training, weights, comp, accumulation, \
                     depth_median, depth_expected, scratch)
#define LAUNCH_R_C(I) \
  do { if (C == 1) LAUNCH_R(I, 1); else if (C == 3) LAUNCH_R(I, 3); else LAUNCH_R(I, 4); } while (0)
  if (S <= 64) LAUNCH_R_C(1);
  else if (S <= 128) LAUNCH_R_C(2);
  else LAUNCH_R_C(4);
#undef LAUNCH_R_C
#undef LAUNCH_R
  TN_CHECK_LAUNCH("tn_render_fwd");
  if (clip_nblk) *clip_nblk = nblk;
  if (depth_expected != nullptr && launch_clip) {
    hipLaunchKernelGGL(k_clip_depth_blocks, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, 512 * 8), 1024)), dim3(512), 0, st, depth_expected, scratch, nblk, RENDER_MAX_BLOCKS, N);
    TN_CHECK_LAUNCH("tn_render_fwd(clip)");
  }
  return TN_OK;
}
extern "C" int tn_render_fwd(const float* e_bins, const float* density, const float* rgb, int64_t N, int32_t S, int32_t C, int32_t training,
                             float* weights, float* comp, float* accumulation, float* depth_median, float* depth_expected, float* scratch,
                             tn_stream_t stream) {
  return tn_render_fwd_ex(e_bins, density, rgb, N, S, C, training, weights, comp, accumulation, depth_median, depth_expected, scratch, true, nullptr, stream);
}

extern "C" int tn_composite_fwd(const float* rgb, const float* weights, const float* e_bins, int64_t N, int32_t S, int32_t C, int32_t training,
                                float* comp, float* accumulation, float* depth_median, float* depth_expected, uint32_t* steps_minmax,
                                tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(rgb && weights && e_bins && comp, "tn_composite_fwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_composite_fwd: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(C == 1 || C == 3 || C == 4, "tn_composite_fwd: unsupported channel count %d", C);
  if (N == 0) return TN_OK;
  dim3 grid((unsigned)std::min<int64_t>(tn_cdiv(N, RAYS_PER_BLOCK), 512)), block(BLOCK);
#define LAUNCH_COMP(I, CC)                                                                                                                \
  hipLaunchKernelGGL((k_composite_fwd<I, CC>), grid, block, 0, tn_s(stream), rgb, weights, e_bins, N, S, training, comp, accumulation, \
                     depth_median, depth_expected, steps_minmax)
#define LAUNCH_COMP_C(I) \
  do { if (C == 1) LAUNCH_COMP(I, 1); else if (C == 3) LAUNCH_COMP(I, 3); else LAUNCH_COMP(I, 4); } while (0)
  if (S <= 64) LAUNCH_COMP_C(1);
  else if (S <= 128) LAUNCH_COMP_C(2);
  else LAUNCH_COMP_C(4);
#undef LAUNCH_COMP_C
#undef LAUNCH_COMP
  TN_CHECK_LAUNCH("tn_composite_fwd");
  return TN_OK;
}

__global__ void k_clip_depth(float* __restrict__ d, const uint32_t* __restrict__ mm, int64_t N) {
  float lo = __uint_as_float(mm[0]), hi = __uint_as_float(mm[1]);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    float v = d[i];
    // torch.clip(x, lo, hi) == min(max(x, lo), hi); NaN propagates
    if (v == v) v = fminf(fmaxf(v, lo), hi);
    d[i] = v;
  }
}
extern "C" int tn_clip_depth(float* depth_expected, const uint32_t* steps_minmax, int64_t N, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(depth_expected && steps_minmax && N >= 0, "tn_clip_depth: bad argument");
  if (N == 0) return TN_OK;
  hipLaunchKernelGGL(k_clip_depth, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, 256), 1024)), dim3(256), 0, tn_s(stream), depth_expected,
                     steps_minmax, N);
  TN_CHECK_LAUNCH("tn_clip_depth");
  return TN_OK;
}

// comp_c = sum_s w_s rgb_sc + rgb_{S-1,c} (1 - sum_s w_s)
//   d rgb_sc = w_s g_c  (+ (1 - sum w) g_c for s = S-1) ;  d w_s = sum_c g_c (rgb_sc - rgb_{S-1,c})
template <int ITEMS, int C>
__global__ void __launch_bounds__(BLOCK) k_composite_bwd(const float* __restrict__ rgb, const float* __restrict__ weights,
                                                         const float* __restrict__ d_comp, int64_t N, int S, float* __restrict__ d_rgb,
                                                         float* __restrict__ d_weights) {
  int lane = tn_lane();
  int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
  if (ray >= N) return;
  float g[C], last[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { g[c] = d_comp[ray * C + c]; last[c] = rgb[(ray * S + (S - 1)) * C + c]; }
  float wsum = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    if (i < S) wsum += weights[ray * S + i];
  }
  wsum = tn_wave_sum(wsum);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    if (i < S) {
      float w = weights[ray * S + i];
      float dw = 0.0f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float v = rgb[(ray * S + i) * C + c];
        dw += g[c] * (v - last[c]);
        float dr = w * g[c];
        if (i == S - 1) dr += (1.0f - wsum) * g[c];
        d_rgb[(ray * S + i) * C + c] = dr;
      }
      d_weights[ray * S + i] += dw;
    }
  }
}

extern "C" int tn_composite_bwd(const float* rgb, const float* weights, const float* d_comp, int64_t N, int32_t S, int32_t C, float* d_rgb,
                                float* d_weights, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(rgb && weights && d_comp && d_rgb && d_weights, "tn_composite_bwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_composite_bwd: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(C == 1 || C == 3 || C == 4, "tn_composite_bwd: unsupported channel count %d", C);
  if (N == 0) return TN_OK;
  dim3 grid((unsigned)tn_cdiv(N, RAYS_PER_BLOCK)), block(BLOCK);
#define LAUNCH_CB(I, CC) hipLaunchKernelGGL((k_composite_bwd<I, CC>), grid, block, 0, tn_s(stream), rgb, weights, d_comp, N, S, d_rgb, d_weights)
#define LAUNCH_CB_C(I) \
  do { if (C == 1) LAUNCH_CB(I, 1); else if (C == 3) LAUNCH_CB(I, 3); else LAUNCH_CB(I, 4); } while (0)
  if (S <= 64) LAUNCH_CB_C(1);
  else if (S <= 128) LAUNCH_CB_C(2);
  else LAUNCH_CB_C(4);
#undef LAUNCH_CB_C
#undef LAUNCH_CB
  TN_CHECK_LAUNCH("tn_composite_bwd");
  return TN_OK;
}

// backward of tn_render_fwd in one launch: composite_bwd, then weights_bwd on d_weights_in + the compositing term (held in registers;
// d_weights_in is NOT updated).  Same arithmetic, same order as tn_composite_bwd followed by tn_weights_bwd.
// one ray: wk = the ray's weights (lane * ITEMS + k; 0 beyond S), g = d composite, gw_in = d weights of the ray as handed in (any address space);
// st / en / dn / rv / lastc = the ray's data (ray_load_bins, ray_load_rgb)
template <int ITEMS, int C>
__device__ __forceinline__ void render_bwd_compute(const float (&st)[ITEMS], const float (&en)[ITEMS], const float (&dn)[ITEMS], const float (&rv)[ITEMS][C],
                                                   const float (&last)[C], const float (&wk)[ITEMS], const float (&g)[C], const float* gw_in, int64_t ray,
                                                   int S, float* __restrict__ d_rgb, float* __restrict__ d_density, int lane) {
  float dd[ITEMS], delta[ITEMS], gw[ITEMS];
  float wsum = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    if (i < S) wsum += wk[k];
  }
  wsum = tn_wave_sum(wsum);
  double loc = 0.0;
  float sloc = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    dd[k] = delta[k] = gw[k] = 0.0f;
    if (i < S) {
      float dw = 0.0f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float v = rv[k][c];
        dw += g[c] * (v - last[c]);
        float dr = wk[k] * g[c];
        if (i == S - 1) dr += (1.0f - wsum) * g[c];
        d_rgb[(ray * S + i) * C + c] = dr;
      }
      gw[k] = gw_in[i] + dw;
      delta[k] = en[k] - st[k];
      dd[k] = delta[k] * dn[k];
    }
    loc += (double)dd[k];
    sloc += gw[k] * wk[k];
  }
  double incl = tn_wave_incl_scan_d(loc, lane);
  double run = tn_excl_from_incl_d(incl, lane);
  float sincl = tn_wave_incl_rscan(sloc, lane);
  float suffix = tn_rexcl_from_incl(sincl, lane);
  float T[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) { T[k] = expf(-(float)run); run += (double)dd[k]; }
#pragma unroll
  for (int k = ITEMS - 1; k >= 0; --k) {
    int i = lane * ITEMS + k;
    float gg = gw[k] * expf(-dd[k]) * T[k] - suffix;
    if (i < S) d_density[ray * S + i] = gg * delta[k];
    suffix += gw[k] * wk[k];
  }
}
template <int ITEMS, int C>
__device__ __forceinline__ void render_bwd_ray(const float* __restrict__ e_bins, const float* __restrict__ density, const float* __restrict__ rgb,
                                               const float (&wk)[ITEMS], const float (&g)[C], const float* gw_in, int64_t ray, int S,
                                               float* __restrict__ d_rgb, float* __restrict__ d_density, int lane) {
  float st[ITEMS], en[ITEMS], dn[ITEMS], rv[ITEMS][C], lastc[C];
  ray_load_bins<ITEMS>(e_bins, density, S, ray, lane, st, en, dn);
  ray_load_rgb<ITEMS, C>(rgb, S, ray, lane, rv, lastc);
  render_bwd_compute<ITEMS, C>(st, en, dn, rv, lastc, wk, g, gw_in, ray, S, d_rgb, d_density, lane);
}
template <int ITEMS, int C>
__global__ void __launch_bounds__(BLOCK) k_render_bwd(const float* __restrict__ e_bins, const float* __restrict__ density,
                                                      const float* __restrict__ rgb, const float* __restrict__ weights,
                                                      const float* __restrict__ d_comp, const float* __restrict__ d_weights_in, int64_t N, int S,
                                                      float* __restrict__ d_rgb, float* __restrict__ d_density) {
  int lane = tn_lane();
  int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
  if (ray >= N) return;
  float g[C], wk[ITEMS];
#pragma unroll
  for (int c = 0; c < C; ++c) g[c] = d_comp[ray * C + c];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    wk[k] = (i < S) ? weights[ray * S + i] : 0.0f;
  }
  render_bwd_ray<ITEMS, C>(e_bins, density, rgb, wk, g, d_weights_in + ray * S, ray, S, d_rgb, d_density, lane);
}

extern "C" int tn_render_bwd(const float* e_bins, const float* density, const float* rgb, const float* weights, const float* d_comp,
                             const float* d_weights_in, int64_t N, int32_t S, int32_t C, float* d_rgb, float* d_density, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(e_bins && density && rgb && weights && d_comp && d_weights_in && d_rgb && d_density, "tn_render_bwd: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_render_bwd: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(C == 1 || C == 3 || C == 4, "tn_render_bwd: unsupported channel count %d", C);
  dim3 grid((unsigned)tn_cdiv(N, RAYS_PER_BLOCK)), block(BLOCK);
#define LAUNCH_RB(I, CC) \
  hipLaunchKernelGGL((k_render_bwd<I, CC>), grid, block, 0, tn_s(stream), e_bins, density, rgb, weights, d_comp, d_weights_in, N, S, d_rgb, d_density)
#define LAUNCH_RB_C(I) \
  do { if (C == 1) LAUNCH_RB(I, 1); else if (C == 3) LAUNCH_RB(I, 3); else LAUNCH_RB(I, 4); } while (0)
  if (S <= 64) LAUNCH_RB_C(1);
  else if (S <= 128) LAUNCH_RB_C(2);
  else LAUNCH_RB_C(4);
#undef LAUNCH_RB_C
#undef LAUNCH_RB
  TN_CHECK_LAUNCH("tn_render_bwd");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ distortion loss
// per ray: L = sum_ij w_i w_j |m_i - m_j| + (1/3) sum_i w_i^2 (t_{i+1}-t_i),  m = bin centres in s-space
//   dL/dw_i = 2 sum_j w_j |m_i - m_j| + (2/3) w_i (t_{i+1}-t_i)      (s-space bins carry no gradient)
// loss_out += mult * mean_over_rays(L)
// LDS of the loss bodies: ONE buffer per block, carved up by whichever body the block runs (a block of the fused launch runs exactly one),
// small enough for 8 blocks per CU -- the whole grid of tn_train_losses is resident at once.
#define LOSS_SMEM_BYTES (RAYS_PER_BLOCK * 3600 + 16)
// one ray; returns the lane's share of L.  tlo / thi / dwf: s_bins[i], s_bins[i + 1] and the handed-in d weights for i = lane + 64 k; the ray's
// weights are in sh_w.  d_w_row (may be NULL) receives dwf + this term, gw_out (LDS row, may be NULL) the same values.
template <int DK>
__device__ __forceinline__ float distortion_compute(const float (&tlo)[DK], const float (&thi)[DK], const float (&dwf)[DK], int S, float scale,
                                                    float* __restrict__ d_w_row, float* gw_out, float* sh_w, float* sh_m, int lane) {
#pragma unroll
  for (int k = 0; k < DK; ++k) {
    const int i = lane + 64 * k;
    if (i < S) sh_m[i] = (thi[k] + tlo[k]) / 2.0f;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float total = 0.0f;
#pragma unroll
  for (int k = 0; k < DK; ++k) {
    const int i = lane + 64 * k;
    if (i < S) {
      float wi = sh_w[i], mi = sh_m[i];
      float inner = 0.0f;
      for (int j = 0; j < S; ++j) inner += sh_w[j] * fabsf(mi - sh_m[j]);
      float width = thi[k] - tlo[k];
      total += wi * inner + wi * wi * width / 3.0f;
      if (d_w_row != nullptr) {
        const float v = dwf[k] + scale * (2.0f * inner + 2.0f * wi * width / 3.0f);
        d_w_row[i] = v;
        if (gw_out != nullptr) gw_out[i] = v;
      }
    }
  }
  __threadfence_block();
  return total;
}
__device__ __forceinline__ float distortion_ray(const float* __restrict__ t, const float* __restrict__ w_row, int S, float scale,
                                                float* __restrict__ d_w_row, float* sh_w, float* sh_m, int lane) {
  constexpr int DK = TN_MAX_SAMPLES / 64;
  float tlo[DK], thi[DK], dwf[DK];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < DK; ++k) {
    const int i = lane + 64 * k;
    tlo[k] = thi[k] = dwf[k] = 0.0f;
    if (i < S) {
      sh_w[i] = w_row[i];
      tlo[k] = t[i]; thi[k] = t[i + 1];
      if (d_w_row != nullptr) dwf[k] = d_w_row[i];
    }
  }
  return distortion_compute<DK>(tlo, thi, dwf, S, scale, d_w_row, nullptr, sh_w, sh_m, lane);
}
__device__ __forceinline__ void distortion_body(const float* __restrict__ s_bins, const float* __restrict__ weights, int64_t N, int S, float mult,
                                                float* __restrict__ loss_out, float* __restrict__ d_weights, unsigned char* smem) {
  float (*sh_w)[TN_MAX_SAMPLES] = reinterpret_cast<float (*)[TN_MAX_SAMPLES]>(smem);
  float (*sh_m)[TN_MAX_SAMPLES] = reinterpret_cast<float (*)[TN_MAX_SAMPLES]>(smem + RAYS_PER_BLOCK * TN_MAX_SAMPLES * 4);
  float* sh_part = reinterpret_cast<float*>(smem + RAYS_PER_BLOCK * 3600);
  static_assert(2 * TN_MAX_SAMPLES * 4 <= 3600, "loss smem");
  int lane = tn_lane();
  int wv = threadIdx.x >> 6;
  float wave_total = 0.0f;
  float scale = mult / (float)N;
  for (int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv; ray < N; ray += (int64_t)gridDim.x * RAYS_PER_BLOCK)
    wave_total += tn_wave_sum(distortion_ray(s_bins + ray * (S + 1), weights + ray * S, S, scale, d_weights ? d_weights + ray * S : nullptr, sh_w[wv], sh_m[wv], lane));
  if (lane == 0) sh_part[wv] = wave_total;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f;
    for (int w = 0; w < RAYS_PER_BLOCK; ++w) a += sh_part[w];
    if (a != 0.0f) atomicAdd(loss_out, a * scale);
  }
}

__global__ void __launch_bounds__(BLOCK) k_distortion(const float* __restrict__ s_bins, const float* __restrict__ weights, int64_t N, int S,
                                                      float mult, float* __restrict__ loss_out, float* __restrict__ d_weights) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LOSS_SMEM_BYTES];
  distortion_body(s_bins, weights, N, S, mult, loss_out, d_weights, smem);
}

extern "C" int tn_distortion_loss(const float* s_bins, const float* weights, int64_t N, int32_t S, float mult, float* loss_out, float* d_weights,
                                  tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(s_bins && weights && loss_out, "tn_distortion_loss: null pointer");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_distortion_loss: bad N=%lld S=%d", (long long)N, S);
  if (N == 0) return TN_OK;
  hipLaunchKernelGGL(k_distortion, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, RAYS_PER_BLOCK), 512)), dim3(BLOCK), 0, tn_s(stream), s_bins, weights, N, S, mult,
                     loss_out, d_weights);
  TN_CHECK_LAUNCH("tn_distortion_loss");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ interlevel loss
// c,w: fine level (S_f bins+1 / weights, detached);  cp,wp: proposal level.
//   w_outer_i = cy[hi_i + 1] - cy[lo_i],  cy = [0, cumsum(wp)],
//   lo_i = clamp(searchsorted_right(cp[:-1], c_i) - 1, 0, Sp-1), hi_i = clamp(searchsorted_right(cp[1:], c_{i+1}), 0, Sp-1)
//   loss = mean_{rays, i} clip(w_i - w_outer_i, 0)^2 / (w_i + 1e-7)
//   d wp_k = sum_i ([lo_i <= k <= hi_i] - [hi_i < k < lo_i]) * g_i,   g_i = -2 clip(w_i - w_outer_i,0) / (w_i + eps) / (N*S_f)
// The gradient is summed directly over the covering fine intervals (all g_i have one sign): no difference-array / prefix-sum, whose
// cancellation residue (1e-16) would become full-size Adam steps on table entries whose true gradient is exactly 0.
// per wave: cp[257] cy[257] g[256] floats, lo[256] hi[256] bytes (proposal-bin indices < 256) = 3592 bytes
struct InterlevelLds { float cp[TN_MAX_SAMPLES + 1]; float cy[TN_MAX_SAMPLES + 1]; float g[TN_MAX_SAMPLES]; uint8_t lo[TN_MAX_SAMPLES]; uint8_t hi[TN_MAX_SAMPLES]; };
static_assert(sizeof(InterlevelLds) <= 3600 && TN_MAX_SAMPLES <= 256, "loss smem");
// one ray: c / wf = the fine level's s-space bins and weights (wf in any address space), cp / wp = the proposal level's, d_wp = its d weights
// (accumulated; may be NULL).  Returns the lane's share of the loss sum.
__device__ __forceinline__ float interlevel_ray(const float* __restrict__ c, const float* wf, int Sf, const float* __restrict__ cp,
                                                const float* __restrict__ wp, int Sp, float scale, float* __restrict__ d_wp, InterlevelLds& wl,
                                                int lane) {
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // every load of the ray first: the stages below each start with data nobody has touched yet, and written where they are used each is a
  // round trip of its own (the proposal bins' copy into LDS alone was five in a row)
  constexpr int DK = TN_MAX_SAMPLES / 64;
  const int ITEMS = 4;
  float v[ITEMS], cpv[DK + 1], t0v[DK], t1v[DK], wfv[DK], dwv[DK];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    v[k] = (i < Sp) ? wp[i] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k <= DK; ++k) {
    const int i = lane + 64 * k;
    cpv[k] = (i <= Sp) ? cp[i] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < DK; ++k) {
    const int i = lane + 64 * k;
    t0v[k] = t1v[k] = wfv[k] = dwv[k] = 0.0f;
    if (i < Sf) { t0v[k] = c[i]; t1v[k] = c[i + 1]; wfv[k] = wf[i]; }
    if (d_wp != nullptr && i < Sp) dwv[k] = d_wp[i];
  }
  // cumsum of wp (torch CPU: double accumulate, float per element); lane owns 4 contiguous entries (Sp<=256)
  double loc = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) loc += (double)v[k];
  double incl = tn_wave_incl_scan_d(loc, lane);
  double run = tn_excl_from_incl_d(incl, lane);
  if (lane == 0) wl.cy[0] = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    run += (double)v[k];
    if (i < Sp) wl.cy[i + 1] = (float)run;
  }
#pragma unroll
  for (int k = 0; k <= DK; ++k)
    if (lane + 64 * k <= Sp) wl.cp[lane + 64 * k] = cpv[k];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float total = 0.0f;
#pragma unroll
  for (int kk = 0; kk < DK; ++kk) {
    const int i = lane + 64 * kk;
    if (i >= Sf) continue;
    float t0 = t0v[kk], t1 = t1v[kk];
    int lo = 0, hi = Sp;  // searchsorted right over the starts cp[0..Sp-1]
    while (lo < hi) { int m = (lo + hi) >> 1; if (wl.cp[m] <= t0) lo = m + 1; else hi = m; }
    int ilo = lo - 1; ilo = ilo < 0 ? 0 : (ilo > Sp - 1 ? Sp - 1 : ilo);
    lo = 0; hi = Sp;      // searchsorted right over the ends cp[1..Sp]
    while (lo < hi) { int m = (lo + hi) >> 1; if (wl.cp[m + 1] <= t1) lo = m + 1; else hi = m; }
    int ihi = lo > Sp - 1 ? Sp - 1 : lo;
    float w_outer = wl.cy[ihi + 1] - wl.cy[ilo];
    float w = wfv[kk];
    float d = w - w_outer;
    if (d < 0.0f) d = 0.0f;
    total += d * d / (w + 1.0e-7f);
    wl.g[i] = -2.0f * d / (w + 1.0e-7f) * scale;
    wl.lo[i] = (uint8_t)ilo;
    wl.hi[i] = (uint8_t)ihi;
  }
  if (d_wp != nullptr) {
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // lo_i and hi_i do not decrease with i (the fine bins are sorted), so the fine intervals that cover proposal bin k are ONE run
    // [B, A): A = #{i : lo_i <= k}, B = #{i : hi_i < k}; should B exceed A, [A, B) is the run of inverted intervals (hi_i < k < lo_i) that the
    // reference's formula subtracts.  Two binary searches and a short in-order sum per lane -- the same additions in the same order as a walk
    // over every interval (48 dependent LDS round trips per 64 bins before).
    bool mono = true;
    for (int i = lane; i < Sf; i += 64)
      if (i > 0 && (wl.lo[i] < wl.lo[i - 1] || wl.hi[i] < wl.hi[i - 1])) mono = false;
    if (__all(mono)) {
#pragma unroll
      for (int kk = 0; kk < DK; ++kk) {
        const int k = lane + 64 * kk;
        if (k >= Sp) continue;
        int l = 0, h = Sf;
        while (l < h) { int m = (l + h) >> 1; if (wl.lo[m] <= k) l = m + 1; else h = m; }
        const int A = l;
        l = 0; h = Sf;
        while (l < h) { int m = (l + h) >> 1; if (wl.hi[m] < k) l = m + 1; else h = m; }
        const int B = l;
        float acc = 0.0f;
        if (B < A) { for (int i = B; i < A; ++i) acc += wl.g[i]; }
        else { for (int i = A; i < B; ++i) acc -= wl.g[i]; }
        if (acc != 0.0f) d_wp[k] = dwv[kk] + acc;
      }
    } else {
    // unsorted fine bins (never from the samplers): walk every interval with a non-zero g_i (wave-uniform bit masks, 64 intervals per word)
    for (int k0 = 0; k0 < Sp; k0 += 64) {
      int k = k0 + lane;
      float acc = 0.0f;
      for (int i0 = 0; i0 < Sf; i0 += 64) {
        int ii = i0 + lane;
        unsigned long long m = __ballot(ii < Sf && wl.g[ii < Sf ? ii : 0] != 0.0f);
        while (m) {
          int i = i0 + __builtin_ctzll(m);
          m &= m - 1;
          int a = wl.lo[i], b = wl.hi[i];
          float gi = wl.g[i];
          if (a <= k && k <= b) acc += gi;
          else if (b < k && k < a) acc -= gi;
        }
      }
      if (k < Sp && acc != 0.0f) d_wp[k] += acc;
    }
    }
  }
  return total;
}
__device__ __forceinline__ void interlevel_body(const float* __restrict__ c_bins, const float* __restrict__ w_fine, int Sf,
                                                const float* __restrict__ p_bins, const float* __restrict__ w_prop, int Sp, int64_t N, float mult,
                                                float* __restrict__ loss_out, float* __restrict__ d_w_prop, unsigned char* smem) {
  int lane = tn_lane();
  int wv = threadIdx.x >> 6;
  InterlevelLds& wl = *reinterpret_cast<InterlevelLds*>(smem + wv * 3600);
  float* sh_part = reinterpret_cast<float*>(smem + RAYS_PER_BLOCK * 3600);
  float wave_total = 0.0f;
  float scale = mult / ((float)N * (float)Sf);
  for (int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv; ray < N; ray += (int64_t)gridDim.x * RAYS_PER_BLOCK)
    wave_total += tn_wave_sum(interlevel_ray(c_bins + ray * (Sf + 1), w_fine + ray * Sf, Sf, p_bins + ray * (Sp + 1), w_prop + ray * Sp, Sp, scale,
                                             d_w_prop ? d_w_prop + ray * Sp : nullptr, wl, lane));
  if (lane == 0) sh_part[wv] = wave_total;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f;
    for (int w = 0; w < RAYS_PER_BLOCK; ++w) a += sh_part[w];
    if (a != 0.0f) atomicAdd(loss_out, a * scale);
  }
}

__global__ void __launch_bounds__(BLOCK) k_interlevel(const float* __restrict__ c_bins, const float* __restrict__ w_fine, int Sf,
                                                      const float* __restrict__ p_bins, const float* __restrict__ w_prop, int Sp, int64_t N,
                                                      float mult, float* __restrict__ loss_out, float* __restrict__ d_w_prop) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LOSS_SMEM_BYTES];
  interlevel_body(c_bins, w_fine, Sf, p_bins, w_prop, Sp, N, mult, loss_out, d_w_prop, smem);
}

// K7: distortion + every interlevel term of one branch in ONE launch -- blockIdx.y = 0 is the distortion loss of the fine level,
// blockIdx.y = 1 + i the interlevel loss against proposal level i.  The three are independent (different outputs), each ~15 us: one launch
// runs them side by side instead of back to back.
struct PropLossArgs {
  const float* s_bins_fine;
  const float* w_fine;
  int Sf;
  int num_props;
  const float* s_bins_prop[TN_MAX_PROP_LEVELS];
  const float* w_prop[TN_MAX_PROP_LEVELS];
  float* d_w_prop[TN_MAX_PROP_LEVELS];
  int Sp[TN_MAX_PROP_LEVELS];
  int64_t N;
  float distortion_mult, interlevel_mult;
  float* distortion_out;
  float* interlevel_out;
  float* d_w_fine;
  // optional first slice (blockIdx.y = 0): the pixel terms of the same iteration (tn_train_losses); pred_rgb == NULL: none
  const float* pred_rgb; const float* pred_th; const float* image; const float* is_thermal;
  int rs, ts, pixel_blocks;
  float thermal_mult, tv_mult, cross_mult;
  float* pixel_losses; float* d_pred_rgb; float* d_pred_th;
  float* loss_lines;  // NULL: every term adds into its own output; else [TN_LOSS_LINES][16], see k_proposal_losses
  // co-work of the pixel slice's idle blocks (tn_train_losses_clip): the batch-wide clip of the expected depth the renderer launch before this
  // one left undone; clip_d == NULL: none
  float* clip_d; const float* clip_scratch; int clip_nblk, clip_hi_off, clip_blocks;
};
__global__ void __launch_bounds__(BLOCK, 6) k_proposal_losses(PropLossArgs a) {
  // Every block ends with one float atomic per loss term, and atomics into ONE 64-byte line execute one after the other (~25 ns each):
  // 512 blocks x 3 slices into the same line is ~40 us, more than the losses themselves take.  With loss_lines the sums are spread over
  // TN_LOSS_LINES lines (block b adds into line b % TN_LOSS_LINES, slot = the term's index) and tn_losses_finish adds the lines up.
  float* line = a.loss_lines ? a.loss_lines + 16 * (blockIdx.x & (TN_LOSS_LINES - 1)) : nullptr;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LOSS_SMEM_BYTES];
  // slice order = dispatch order: the pixel terms first (a handful of blocks with the longest dependent chain: they must not wait for a
  // second round of blocks), then distortion, then the interlevel terms
  int slice = blockIdx.y;
  if (a.pred_rgb != nullptr) {
    if (slice == 0) {
      if ((int)blockIdx.x >= a.pixel_blocks) {  // whole block leaves together
        if (a.clip_d != nullptr && (int)blockIdx.x < a.pixel_blocks + a.clip_blocks)
          clip_depth_body(a.clip_d, a.clip_scratch, a.clip_nblk, a.clip_hi_off, a.N, (int)blockIdx.x - a.pixel_blocks, a.clip_blocks,
                          reinterpret_cast<float*>(smem), reinterpret_cast<float*>(smem) + 8);
        return;
      }
      pixel_losses_body(a.pred_rgb, a.rs, a.pred_th, a.ts, a.image, a.is_thermal, a.N, a.thermal_mult, a.tv_mult, a.cross_mult,
                        line ? line : a.pixel_losses, a.d_pred_rgb, a.d_pred_th, blockIdx.x, a.pixel_blocks);
      return;
    }
    --slice;
  }
  if (slice == 0) {
    distortion_body(a.s_bins_fine, a.w_fine, a.N, a.Sf, a.distortion_mult, line ? line + 9 : a.distortion_out, a.d_w_fine, smem);
  } else {
    int i = slice - 1;
    interlevel_body(a.s_bins_fine, a.w_fine, a.Sf, a.s_bins_prop[i], a.w_prop[i], a.Sp[i], a.N, a.interlevel_mult, line ? line + 8 : a.interlevel_out,
                    a.d_w_prop[i], smem);
  }
}

extern "C" int tn_interlevel_loss(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, const float* s_bins_prop,
                                  const float* weights_prop, int32_t S_prop, int64_t N, float mult, float* loss_out, float* d_weights_prop,
                                  tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(s_bins_fine && weights_fine && s_bins_prop && weights_prop && loss_out, "tn_interlevel_loss: null pointer");
  TN_REQUIRE(N >= 0 && S_fine >= 1 && S_fine <= TN_MAX_SAMPLES && S_prop >= 1 && S_prop <= TN_MAX_SAMPLES,
             "tn_interlevel_loss: bad N=%lld S_fine=%d S_prop=%d", (long long)N, S_fine, S_prop);
  if (N == 0) return TN_OK;
  hipLaunchKernelGGL(k_interlevel, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, RAYS_PER_BLOCK), 512)), dim3(BLOCK), 0, tn_s(stream), s_bins_fine, weights_fine, S_fine,
                     s_bins_prop, weights_prop, S_prop, N, mult, loss_out, d_weights_prop);
  TN_CHECK_LAUNCH("tn_interlevel_loss");
  return TN_OK;
}

static int launch_train_losses(const char* who, const float* s_bins_fine, const float* weights_fine, int32_t S_fine, int32_t num_props,
                               const float* const* s_bins_prop, const float* const* weights_prop, const int32_t* S_prop,
                               float* const* d_weights_prop, int64_t N, float distortion_mult, float interlevel_mult, float* distortion_out,
                               float* interlevel_out, float* d_weights_fine, const float* pred_rgb, int32_t rgb_stride, const float* pred_thermal,
                               int32_t thermal_stride, const float* image, const float* is_thermal, float thermal_mult, float tv_mult,
                               float cross_mult, float* pixel_losses_out, float* d_pred_rgb, float* d_pred_thermal, float* loss_lines,
                               tn_stream_t stream, float* clip_d = nullptr, const float* clip_scratch = nullptr, int clip_nblk = 0) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(s_bins_fine && weights_fine && ((distortion_out && interlevel_out) || loss_lines), "%s: null pointer", who);
  TN_REQUIRE(N > 0 && S_fine >= 1 && S_fine <= TN_MAX_SAMPLES, "%s: bad N=%lld S_fine=%d", who, (long long)N, S_fine);
  TN_REQUIRE(num_props >= 0 && num_props <= TN_MAX_PROP_LEVELS && (num_props == 0 || (s_bins_prop && weights_prop && S_prop && d_weights_prop)),
             "%s: bad proposal level list (num_props=%d, at most %d)", who, num_props, TN_MAX_PROP_LEVELS);
  PropLossArgs a{};
  a.s_bins_fine = s_bins_fine; a.w_fine = weights_fine; a.Sf = S_fine; a.num_props = num_props; a.N = N;
  a.distortion_mult = distortion_mult; a.interlevel_mult = interlevel_mult;
  a.distortion_out = distortion_out; a.interlevel_out = interlevel_out; a.d_w_fine = d_weights_fine;
  for (int i = 0; i < num_props; ++i) {
    TN_REQUIRE(s_bins_prop[i] && weights_prop[i] && S_prop[i] >= 1 && S_prop[i] <= TN_MAX_SAMPLES, "%s: bad proposal level %d", who, i);
    a.s_bins_prop[i] = s_bins_prop[i]; a.w_prop[i] = weights_prop[i]; a.Sp[i] = S_prop[i]; a.d_w_prop[i] = d_weights_prop[i];
  }
  int slices = 1 + num_props;
  if (pred_rgb != nullptr) {
    TN_REQUIRE(pred_thermal && image && is_thermal && (pixel_losses_out || loss_lines), "%s: null pointer in the pixel terms", who);
    TN_REQUIRE(N % 4 == 0, "%s: N=%lld must be a multiple of 4 (2x2 patches)", who, (long long)N);
    TN_REQUIRE(rgb_stride >= 3 && thermal_stride >= 1, "%s: bad strides", who);
    a.pred_rgb = pred_rgb; a.pred_th = pred_thermal; a.image = image; a.is_thermal = is_thermal; a.rs = rgb_stride; a.ts = thermal_stride;
    a.thermal_mult = thermal_mult; a.tv_mult = tv_mult; a.cross_mult = cross_mult;
    a.pixel_losses = pixel_losses_out; a.d_pred_rgb = d_pred_rgb; a.d_pred_th = d_pred_thermal;
    a.pixel_blocks = pixel_loss_blocks(N);
    ++slices;
  }
  a.loss_lines = loss_lines;
  dim3 grid((unsigned)std::min<int64_t>(tn_cdiv(N, RAYS_PER_BLOCK), 512), slices);
  if (clip_d != nullptr && clip_nblk > 0) {
    TN_REQUIRE(pred_rgb != nullptr && clip_scratch != nullptr, "%s: the depth clip rides in the pixel slice", who);
    const int cb = (int)std::min<int64_t>(tn_cdiv(N, BLOCK * 8), 64);
    if (a.pixel_blocks + cb <= (int)grid.x) {
      a.clip_d = clip_d; a.clip_scratch = clip_scratch; a.clip_nblk = clip_nblk; a.clip_hi_off = RENDER_MAX_BLOCKS; a.clip_blocks = cb;
    } else {  // (tiny batches: no idle blocks in the slice)
      hipLaunchKernelGGL(k_clip_depth_blocks, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, 512 * 8), 1024)), dim3(512), 0, tn_s(stream), clip_d, clip_scratch, clip_nblk,
                         RENDER_MAX_BLOCKS, N);
      TN_CHECK_LAUNCH(who);
    }
  }
  hipLaunchKernelGGL(k_proposal_losses, grid, dim3(BLOCK), 0, tn_s(stream), a);
  TN_CHECK_LAUNCH(who);
  return TN_OK;
}

extern "C" int tn_proposal_losses(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, int32_t num_props,
                                  const float* const* s_bins_prop, const float* const* weights_prop, const int32_t* S_prop,
                                  float* const* d_weights_prop, int64_t N, float distortion_mult, float interlevel_mult, float* distortion_out,
                                  float* interlevel_out, float* d_weights_fine, tn_stream_t stream) {
  return launch_train_losses("tn_proposal_losses", s_bins_fine, weights_fine, S_fine, num_props, s_bins_prop, weights_prop, S_prop, d_weights_prop, N,
                             distortion_mult, interlevel_mult, distortion_out, interlevel_out, d_weights_fine, nullptr, 0, nullptr, 0, nullptr, nullptr,
                             0.f, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int tn_train_losses(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, int32_t num_props,
                               const float* const* s_bins_prop, const float* const* weights_prop, const int32_t* S_prop,
                               float* const* d_weights_prop, int64_t N, float distortion_mult, float interlevel_mult, float* d_weights_fine,
                               const float* pred_rgb, int32_t rgb_stride, const float* pred_thermal, int32_t thermal_stride, const float* image,
                               const float* is_thermal, float thermal_mult, float tv_mult, float cross_mult, float* d_pred_rgb,
                               float* d_pred_thermal, float* loss_lines, tn_stream_t stream) {
  TN_REQUIRE(loss_lines != nullptr, "tn_train_losses: null pointer (loss_lines)");
  return launch_train_losses("tn_train_losses", s_bins_fine, weights_fine, S_fine, num_props, s_bins_prop, weights_prop, S_prop, d_weights_prop, N,
                             distortion_mult, interlevel_mult, nullptr, nullptr, d_weights_fine, pred_rgb, rgb_stride, pred_thermal, thermal_stride,
                             image, is_thermal, thermal_mult, tv_mult, cross_mult, nullptr, d_pred_rgb, d_pred_thermal, loss_lines, stream);
}

// ------------------------------------------------------------------------------------------------ renderer + losses + renderer backward
// tn_render_losses_bwd: tn_render_fwd(training) + tn_train_losses + tn_render_bwd of the shared-density model's last level in ONE launch.
// The three are wave-per-ray kernels over the same rays, each latency-bound on its own (10 + 24 + 5 us with two launch gaps between them):
//   blockIdx.y = 0      per 2x2 patch (4 rays = the 4 waves of a block): get_weights + the renderers, then -- composites exchanged through
//                       LDS -- the pixel terms of the patch, the distortion loss of each ray, and the renderers' backward with the weights,
//                       d composite and d weights still in registers / LDS;
//   blockIdx.y = 1 + i  the interlevel loss against proposal level i.  It needs the fine level's weights: recomputed here with the function
//                       slice 0 uses (same inputs, same instructions: the same bits), instead of waiting for slice 0 to publish them.
// Every per-element result is what the three calls give (same expressions in the same order); the loss SUMS are added up in another order
// (per block, then one float atomic per term and block -- as before they differ from run to run in the last bits).
#define RT_SMEM_BYTES (RAYS_PER_BLOCK * 3600 + RAYS_PER_BLOCK * TN_MAX_SAMPLES * 4 + 64)
#ifndef RT_ABLATE
#define RT_ABLATE 0  // (timing experiments: 1 no distortion stage, 2 no pixel stage, 4 no renderer backward, 8 no renderers)
#endif
#define RT_MAX_BLOCKS 2048  // one patch per block up to 8192 rays: every wave renders ONE ray (no second trip on the chain)
static_assert(TN_RENDER_SCRATCH_FLOATS >= 2 * RT_MAX_BLOCKS, "render scratch");
struct RenderTrainArgs {
  const float* e_bins; const float* density; const float* rgb; int64_t N; int S;
  float* weights; float* comp; float* accumulation; float* depth_median; float* depth_expected; float* scratch;
  const float* s_bins_fine; float distortion_mult, interlevel_mult; float* d_w_fine;
  int num_props;
  const float* s_bins_prop[TN_MAX_PROP_LEVELS]; const float* w_prop[TN_MAX_PROP_LEVELS]; float* d_w_prop[TN_MAX_PROP_LEVELS]; int Sp[TN_MAX_PROP_LEVELS];
  const float* image; const float* is_thermal; float thermal_mult, tv_mult, cross_mult; float* d_comp;
  float* loss_lines; float* d_rgb; float* d_density;
};
template <int ITEMS>
__global__ void __launch_bounds__(BLOCK, ITEMS == 1 ? 6 : 3) k_render_train(RenderTrainArgs a) {
  constexpr int C = 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RT_SMEM_BYTES];
  static_assert(3 * RAYS_PER_BLOCK * TN_MAX_SAMPLES * 4 + 64 * 4 <= RT_SMEM_BYTES, "render-train smem");
  const int lane = tn_lane(), wv = threadIdx.x >> 6;
  const int S = a.S;
  const int64_t N = a.N;
  float* line = a.loss_lines + 16 * (blockIdx.x & (TN_LOSS_LINES - 1));
  if (blockIdx.y > 0) {
    const int pi = (int)blockIdx.y - 1, Sp = a.Sp[pi];
    InterlevelLds& wl = *reinterpret_cast<InterlevelLds*>(smem + wv * 3600);
    float* sh_wf = reinterpret_cast<float*>(smem + RAYS_PER_BLOCK * 3600) + wv * TN_MAX_SAMPLES;
    float* sh_part = reinterpret_cast<float*>(smem + RAYS_PER_BLOCK * 3600 + RAYS_PER_BLOCK * TN_MAX_SAMPLES * 4);
    const float scale = a.interlevel_mult / ((float)N * (float)S);
    float wave_total = 0.0f;
    for (int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv; ray < N; ray += (int64_t)gridDim.x * RAYS_PER_BLOCK) {
      float w[ITEMS];
      weights_body<ITEMS>(a.e_bins, a.density, S, ray, nullptr, nullptr, lane, w);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < ITEMS; ++k)
        if (lane * ITEMS + k < S) sh_wf[lane * ITEMS + k] = w[k];
      wave_total += tn_wave_sum(interlevel_ray(a.s_bins_fine + ray * (S + 1), sh_wf, S, a.s_bins_prop[pi] + ray * (Sp + 1), a.w_prop[pi] + ray * Sp, Sp, scale,
                                               a.d_w_prop[pi] ? a.d_w_prop[pi] + ray * Sp : nullptr, wl, lane));
    }
    if (lane == 0) sh_part[wv] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
      for (int w = 0; w < RAYS_PER_BLOCK; ++w) t += sh_part[w];
      if (t != 0.0f) atomicAdd(line + 8, t * scale);
    }
    return;
  }
  float (*sh_w)[TN_MAX_SAMPLES] = reinterpret_cast<float (*)[TN_MAX_SAMPLES]>(smem);
  float (*sh_m)[TN_MAX_SAMPLES] = reinterpret_cast<float (*)[TN_MAX_SAMPLES]>(smem + RAYS_PER_BLOCK * TN_MAX_SAMPLES * 4);
  float (*sh_gw)[TN_MAX_SAMPLES] = reinterpret_cast<float (*)[TN_MAX_SAMPLES]>(smem + 2 * RAYS_PER_BLOCK * TN_MAX_SAMPLES * 4);
  float* sh_comp = reinterpret_cast<float*>(smem + 3 * RAYS_PER_BLOCK * TN_MAX_SAMPLES * 4);  // [4 rays][4 channels]
  float* sh_cnt = sh_comp + 16;                                                               // [4]
  float* sh_red = sh_cnt + 4;                                                                 // [4 waves][8]
  // RGB rays of the batch (the patch terms' mean): every block counts them itself.  Requested here -- four 16-byte loads per thread up to 4096
  // rays --, added up after the first composite (a scalar loop at this point was a chain of its own in front of the rendering)
  float4 th4[4];
  const bool th_vec = (reinterpret_cast<uintptr_t>(a.is_thermal) & 15) == 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i4 = (int64_t)u * BLOCK + threadIdx.x;
    th4[u] = make_float4(1.f, 1.f, 1.f, 1.f);  // (thermal: not counted)
    if (th_vec && i4 < N / 4) th4[u] = reinterpret_cast<const float4*>(a.is_thermal)[i4];
  }
  bool counted = false;
  float blk_mn = INFINITY, blk_mx = 0.0f;
  float l_rgb = 0.f, l_th = 0.f, l_tv = 0.f, l_cc = 0.f, dist_total = 0.0f, n_rgb_rays = 0.0f;
  const float dscale = a.distortion_mult / (float)N;
  // (N is a multiple of 4: the four waves of a block make the same trips, the block barriers inside are reached by all of them)
  for (int64_t ray = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv; ray < N; ray += (int64_t)gridDim.x * RAYS_PER_BLOCK) {
    // ---- every load the ray's chain of stages needs, requested now (each would otherwise be an HBM round trip in front of its stage)
    float st[ITEMS], en[ITEMS], dn[ITEMS], rv[ITEMS][C], lastc[C];
    ray_load_bins<ITEMS>(a.e_bins, a.density, S, ray, lane, st, en, dn);
    ray_load_rgb<ITEMS, C>(a.rgb, S, ray, lane, rv, lastc);
    const int64_t q4 = ray - wv;  // first ray of the 2x2 patch
    float img_l = 0.0f, th_l = 0.0f, dc_l = 0.0f;
    if (lane < 12) img_l = a.image[q4 * 3 + lane];
    if (lane < 4) { th_l = a.is_thermal[q4 + lane]; dc_l = a.d_comp[ray * C + lane]; }
    float tlo[ITEMS], thi[ITEMS], dwf[ITEMS];  // distortion: lane + 64 k order (ITEMS = ceil(S / 64))
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int i = lane + 64 * k;
      tlo[k] = thi[k] = dwf[k] = 0.0f;
      if (i < S) { tlo[k] = a.s_bins_fine[ray * (S + 1) + i]; thi[k] = a.s_bins_fine[ray * (S + 1) + i + 1]; dwf[k] = a.d_w_fine[ray * S + i]; }
    }
    // ---- get_weights + the renderers
    float w[ITEMS], cv[C] = {0.f, 0.f, 0.f, 0.f};
    float mn = 0.f, mx = 0.f;
#if !(RT_ABLATE & 8)
    weights_compute<ITEMS>(st, en, dn, S, ray, a.weights, nullptr, lane, w);
    composite_compute<ITEMS, C>(w, st, en, rv, lastc, S, ray, 1, a.comp, a.accumulation, a.depth_median, a.depth_expected, lane, mn, mx, cv);
#else
    for (int k = 0; k < ITEMS; ++k) w[k] = st[k] + dn[k] + rv[k][0];
#endif
    blk_mn = fminf(blk_mn, tn_wave_min(mn));
    blk_mx = fmaxf(blk_mx, tn_wave_max(mx));
    if (!counted) {
      counted = true;
      float cnt = 0.0f;  // exact: a count of at most 2^24 ones
#pragma unroll
      for (int u = 0; u < 4; ++u)
        cnt += (th4[u].x == 0.0f ? 1.0f : 0.0f) + (th4[u].y == 0.0f ? 1.0f : 0.0f) + (th4[u].z == 0.0f ? 1.0f : 0.0f) + (th4[u].w == 0.0f ? 1.0f : 0.0f);
      for (int64_t i = (th_vec ? (int64_t)16 * BLOCK : 0) + threadIdx.x; i < N; i += BLOCK) cnt += (a.is_thermal[i] == 0.0f) ? 1.0f : 0.0f;
      cnt = tn_wave_sum(cnt);
      if (lane == 0) sh_cnt[wv] = cnt;
    }
    __syncthreads();  // the previous patch's readers are done with sh_comp
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) sh_comp[wv * 4 + c] = cv[c];
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (lane * ITEMS + k < S) sh_w[wv][lane * ITEMS + k] = w[k];
    __syncthreads();  // the patch's four composites (and, first trip, the counts)
    // ---- pixel terms (tn_pixel_loss.h: same expressions) of this wave's ray = pixel wv of the patch; every lane computes them, lane 0 keeps them
    float g[C] = {0.f, 0.f, 0.f, 0.f};
#if !(RT_ABLATE & 2)
    {
      n_rgb_rays = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
      const float n_patches = n_rgb_rays / 4.0f;
      float pt[4], grey[4];
      bool rgb_patch = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float th = __shfl(th_l, k, 64), nt = 1.0f - th;
        if (th != 0.0f) rgb_patch = false;
        float gsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float img = __shfl(img_l, 3 * k + c, 64);
          const float gt = img * nt;
          if (k == wv) {
            const float aa = gt * nt, bb = sh_comp[k * 4 + c] * nt;
            const float df = aa - bb;
            if (lane == 0) l_rgb += df * df;
            g[c] = __shfl(dc_l, c, 64) + (-2.0f * df * nt / ((float)N * 3.0f));
          }
          gsum += gt;
        }
        grey[k] = gsum / 3.0f;
        const float p = sh_comp[k * 4 + 3];
        pt[k] = p;
        if (k == wv) {
          const float gt_t = __shfl(img_l, 3 * k, 64) * th;
          const float aa = gt_t * th, bb = p * th;
          const float df = aa - bb;
          if (lane == 0) l_th += df * df;
          g[3] = __shfl(dc_l, 3, 64) + (a.thermal_mult * (-2.0f * df * th) / (float)N);
        }
      }
      if (rgb_patch && n_patches > 0.0f) {
        const int A[4] = {1, 2, 3, 3}, B[4] = {0, 0, 1, 2};
        float dp[4] = {0.f, 0.f, 0.f, 0.f};
        const float tvw = a.tv_mult * 0.25f / n_patches, ccw = a.cross_mult * 0.25f / n_patches;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dpt = pt[A[e]] - pt[B[e]];
          const float dg = grey[A[e]] - grey[B[e]];
          if (wv == 0 && lane == 0) { l_tv += fabsf(dpt); l_cc += fabsf(dpt - dg); }  // (the patch's sums once, not four times)
          const float sg = tvw * sgn(dpt) + ccw * sgn(dpt - dg);
          dp[A[e]] += sg;
          dp[B[e]] -= sg;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k == wv) g[3] += dp[k];
      }
      if (lane < C) {
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (c == lane) a.d_comp[ray * C + c] = g[c];
      }
    }
#endif
    // ---- distortion loss of the ray: d weights (handed-in accumulator + this term) to memory and to sh_gw
#if !(RT_ABLATE & 1)
    dist_total += tn_wave_sum(distortion_compute<ITEMS>(tlo, thi, dwf, S, dscale, a.d_w_fine + ray * S, sh_gw[wv], sh_w[wv], sh_m[wv], lane));
#endif
    __builtin_amdgcn_wave_barrier();
    // ---- the renderers' backward on the registers of the forward
#if !(RT_ABLATE & 4)
    render_bwd_compute<ITEMS, C>(st, en, dn, rv, lastc, w, g, sh_gw[wv], ray, S, a.d_rgb, a.d_density, lane);
#endif
  }
  if (lane == 0) {
    float* r = sh_red + wv * 8;
    r[0] = l_rgb; r[1] = l_th; r[2] = l_tv; r[3] = l_cc; r[4] = dist_total; r[5] = blk_mn; r[6] = blk_mx; r[7] = n_rgb_rays;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, mn = INFINITY, mx = 0.0f, n_rgb = 0.0f;
    for (int w = 0; w < RAYS_PER_BLOCK; ++w) {
#pragma unroll
      for (int k = 0; k < 5; ++k) t[k] += sh_red[w * 8 + k];
      mn = fminf(mn, sh_red[w * 8 + 5]);
      mx = fmaxf(mx, sh_red[w * 8 + 6]);
      n_rgb = fmaxf(n_rgb, sh_red[w * 8 + 7]);
    }
    if (a.depth_expected != nullptr) { a.scratch[blockIdx.x] = mn; a.scratch[RT_MAX_BLOCKS + blockIdx.x] = mx; }
    const float n_patches = n_rgb / 4.0f;
    if (t[0] != 0.0f) atomicAdd(line + 0, t[0] / ((float)N * 3.0f));
    if (t[1] != 0.0f) atomicAdd(line + 1, a.thermal_mult * t[1] / (float)N);
    if (n_patches > 0.0f) {
      if (t[2] != 0.0f) atomicAdd(line + 2, a.tv_mult * 0.25f * t[2] / n_patches);
      if (t[3] != 0.0f) atomicAdd(line + 3, a.cross_mult * 0.25f * t[3] / n_patches);
    }
    if (t[4] != 0.0f) atomicAdd(line + 9, t[4] * dscale);
    if (blockIdx.x == 0) { line[4] = n_rgb; line[5] = (float)N - n_rgb; }  // ray counts per spectrum (PSNR metrics)
  }
}

extern "C" int tn_render_losses_bwd(const float* e_bins, const float* density, const float* rgb, int64_t N, int32_t S, int32_t C, float* weights,
                                    float* comp, float* accumulation, float* depth_median, float* depth_expected, float* scratch,
                                    const float* s_bins_fine, int32_t num_props, const float* const* s_bins_prop, const float* const* weights_prop,
                                    const int32_t* S_prop, float* const* d_weights_prop, float distortion_mult, float interlevel_mult,
                                    float* d_weights_fine, const float* image, const float* is_thermal, float thermal_mult, float tv_mult,
                                    float cross_mult, float* d_comp, float* loss_lines, float* d_rgb, float* d_density, int32_t clip_depth,
                                    tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(e_bins && density && rgb && weights && comp && s_bins_fine && d_weights_fine && image && is_thermal && d_comp && loss_lines && d_rgb && d_density,
             "tn_render_losses_bwd: null pointer");
  TN_REQUIRE(N > 0 && N % 4 == 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_render_losses_bwd: bad N=%lld (a multiple of 4: 2x2 patches) S=%d", (long long)N, S);
  TN_REQUIRE(C == 4, "tn_render_losses_bwd: the shared-density model's RGB + thermal composite (4 channels), got %d", C);
  TN_REQUIRE(depth_expected == nullptr || scratch != nullptr, "tn_render_losses_bwd: depth_expected needs the scratch buffer (TN_RENDER_SCRATCH_FLOATS)");
  TN_REQUIRE(num_props >= 0 && num_props <= TN_MAX_PROP_LEVELS && (num_props == 0 || (s_bins_prop && weights_prop && S_prop && d_weights_prop)),
             "tn_render_losses_bwd: bad proposal level list (num_props=%d, at most %d)", num_props, TN_MAX_PROP_LEVELS);
  RenderTrainArgs a{};
  a.e_bins = e_bins; a.density = density; a.rgb = rgb; a.N = N; a.S = S; a.weights = weights; a.comp = comp; a.accumulation = accumulation;
  a.depth_median = depth_median; a.depth_expected = depth_expected; a.scratch = scratch; a.s_bins_fine = s_bins_fine;
  a.distortion_mult = distortion_mult; a.interlevel_mult = interlevel_mult; a.d_w_fine = d_weights_fine; a.num_props = num_props;
  for (int i = 0; i < num_props; ++i) {
    TN_REQUIRE(s_bins_prop[i] && weights_prop[i] && S_prop[i] >= 1 && S_prop[i] <= TN_MAX_SAMPLES, "tn_render_losses_bwd: bad proposal level %d", i);
    a.s_bins_prop[i] = s_bins_prop[i]; a.w_prop[i] = weights_prop[i]; a.Sp[i] = S_prop[i]; a.d_w_prop[i] = d_weights_prop[i];
  }
  a.image = image; a.is_thermal = is_thermal; a.thermal_mult = thermal_mult; a.tv_mult = tv_mult; a.cross_mult = cross_mult; a.d_comp = d_comp;
  a.loss_lines = loss_lines; a.d_rgb = d_rgb; a.d_density = d_density;
  const int nblk = (int)std::min<int64_t>(N / RAYS_PER_BLOCK, RT_MAX_BLOCKS);
  dim3 grid((unsigned)nblk, (unsigned)(1 + num_props)), block(BLOCK);
  hipStream_t st = tn_s(stream);
  if (S <= 64) hipLaunchKernelGGL(k_render_train<1>, grid, block, 0, st, a);
  else if (S <= 128) hipLaunchKernelGGL(k_render_train<2>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(k_render_train<4>, grid, block, 0, st, a);
  TN_CHECK_LAUNCH("tn_render_losses_bwd");
  if (depth_expected != nullptr && clip_depth) {
    hipLaunchKernelGGL(k_clip_depth_blocks, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, 512 * 8), 1024)), dim3(512), 0, st, depth_expected, scratch, nblk, RT_MAX_BLOCKS, N);
    TN_CHECK_LAUNCH("tn_render_losses_bwd(clip)");
  }
  return TN_OK;
}

// tn_train_losses + the batch-wide clip of depth_expected (tn_render_fwd_ex(launch_clip = false) before it) as co-work blocks of the same launch
int tn_train_losses_clip(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, int32_t num_props, const float* const* s_bins_prop,
                         const float* const* weights_prop, const int32_t* S_prop, float* const* d_weights_prop, int64_t N, float distortion_mult,
                         float interlevel_mult, float* d_weights_fine, const float* pred_rgb, int32_t rgb_stride, const float* pred_thermal,
                         int32_t thermal_stride, const float* image, const float* is_thermal, float thermal_mult, float tv_mult, float cross_mult,
                         float* d_pred_rgb, float* d_pred_thermal, float* loss_lines, float* depth_expected, const float* scratch, int clip_nblk,
                         tn_stream_t stream) {
  TN_REQUIRE(loss_lines != nullptr, "tn_train_losses: null pointer (loss_lines)");
  return launch_train_losses("tn_train_losses", s_bins_fine, weights_fine, S_fine, num_props, s_bins_prop, weights_prop, S_prop, d_weights_prop, N,
                             distortion_mult, interlevel_mult, nullptr, nullptr, d_weights_fine, pred_rgb, rgb_stride, pred_thermal, thermal_stride,
                             image, is_thermal, thermal_mult, tv_mult, cross_mult, nullptr, d_pred_rgb, d_pred_thermal, loss_lines, stream, depth_expected,
                             scratch, clip_nblk);
}

extern "C" int tn_weights_resample(const float* e_bins_prev, const float* density_prev, const float* s_bins_prev, int32_t S_prev, float anneal,
                                   const float* u_lin, const float* jitter, const float* nears, const float* fars, int64_t N, int32_t S,
                                   float* weights_prev, float* median_prev, float* s_bins, float* e_bins, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(e_bins_prev && density_prev && s_bins_prev && u_lin && nears && fars && weights_prev && s_bins && e_bins, "tn_weights_resample: null pointer");
  TN_REQUIRE(N > 0 && S >= 1 && S <= TN_MAX_SAMPLES && S_prev >= 1 && S_prev <= TN_MAX_SAMPLES, "tn_weights_resample: bad N=%lld S=%d S_prev=%d",
             (long long)N, S, S_prev);
  dim3 grid((unsigned)tn_cdiv(N, RAYS_PER_BLOCK)), block(BLOCK);
#define LAUNCH_WP(I)                                                                                                                         \
  hipLaunchKernelGGL(k_weights_pdf<I>, grid, block, 0, tn_s(stream), e_bins_prev, density_prev, s_bins_prev, S_prev, anneal, u_lin, jitter, nears, \
                     fars, N, S, weights_prev, median_prev, s_bins, e_bins)
  if (S_prev <= 64) LAUNCH_WP(1);
  else if (S_prev <= 128) LAUNCH_WP(2);
  else LAUNCH_WP(4);
#undef LAUNCH_WP
  TN_CHECK_LAUNCH("tn_weights_resample");
  return TN_OK;
}
