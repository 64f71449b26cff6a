// The per-ray stages of the proposal sampler as device functions (one 64-lane wave owns one ray): RaySamples.get_weights (cameras/rays.py:128-150)
// and PDFSampler.generate_ray_samples (model_components/ray_samplers.py:276-372).  Shared by the kernels of tn_sampler.hip and by the
// next-iteration sampling that rides in the optimiser launch (tn_next_sampling.h): one body, so the two paths produce the same bits.
#pragma once
#include "tn_common.h"

// lane l owns samples [l*ITEMS, (l+1)*ITEMS) of its ray.
// The per-ray bodies come in two halves: the LOADS of the ray's data (bins, density, colours) and the arithmetic on registers.  The one-launch
// kernels call them back to back; tn_render_losses_bwd requests everything a ray's chain of stages will need up front (each stage's first
// touch of new data is an HBM round trip of ~2 us -- five of them in a row were most of that kernel) and keeps it for the backward.
template <int ITEMS>
__device__ __forceinline__ void ray_load_bins(const float* __restrict__ e_bins, const float* __restrict__ density, int S, int64_t ray, int lane,
                                              float (&st)[ITEMS], float (&en)[ITEMS], float (&dn)[ITEMS]) {
  const float* eb = e_bins + ray * (S + 1);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    st[k] = en[k] = dn[k] = 0.0f;
    if (i < S) {
      st[k] = eb[i]; en[k] = eb[i + 1];
      if (density != nullptr) dn[k] = density[ray * S + i];
    }
  }
}
template <int ITEMS>
__device__ __forceinline__ void weights_compute(const float (&st)[ITEMS], const float (&en)[ITEMS], const float (&dn)[ITEMS], int S, int64_t ray,
                                                float* __restrict__ weights, float* __restrict__ median_depth, int lane, float (&w)[ITEMS]) {
  float dd[ITEMS], mid[ITEMS];
  double loc = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    dd[k] = 0.0f; mid[k] = 0.0f;
    if (i < S) {
      dd[k] = (en[k] - st[k]) * dn[k];
      mid[k] = (st[k] + en[k]) / 2.0f;
    }
    loc += (double)dd[k];
  }
  double incl = tn_wave_incl_scan_d(loc, lane);
  double run = tn_excl_from_incl_d(incl, lane);  // exclusive prefix over earlier lanes
  double wloc = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    float trans = (float)run;  // cumsum of dd[:-1] in double, rounded to float per element (torch CPU cumsum)
    float a = 1.0f - expf(-dd[k]);
    float T = expf(-trans);
    w[k] = tn_nan_to_num(a * T);
    if (i >= S) w[k] = 0.0f;
    else if (weights != nullptr) weights[ray * S + i] = w[k];
    run += (double)dd[k];
    wloc += (double)w[k];
  }
  if (median_depth != nullptr) {
    // cumsum(weights) (double accumulate -> float), first index with cum >= 0.5 (searchsorted left), clamp, gather midpoints
    double wincl = tn_wave_incl_scan_d(wloc, lane);
    double wrun = tn_excl_from_incl_d(wincl, lane);
    int cnt = 0;  // number of samples with cum < 0.5
    float best = 0.0f;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i = lane * ITEMS + k;
      wrun += (double)w[k];
      if (i < S && (float)wrun < 0.5f) cnt++;
    }
    int total = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    int idx = total < S - 1 ? total : S - 1;
    // gather mid[idx] from its owner lane
    int owner = idx / ITEMS, slot = idx - owner * ITEMS;
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) if (k == slot) v = mid[k];
    best = __shfl(v, owner, 64);
    if (lane == 0) median_depth[ray] = best;
  }
}
template <int ITEMS>
__device__ __forceinline__ void weights_body(const float* __restrict__ e_bins, const float* __restrict__ density, int S, int64_t ray,
                                             float* __restrict__ weights, float* __restrict__ median_depth, int lane, float (&w)[ITEMS]) {
  float st[ITEMS], en[ITEMS], dn[ITEMS];
  ray_load_bins<ITEMS>(e_bins, density, S, ray, lane, st, en, dn);
  weights_compute<ITEMS>(st, en, dn, S, ray, weights, median_depth, lane, w);
}
// w_raw[k] = weight of sample lane*ITEMS + k of the previous level (0 beyond Sp): from memory (k_pdf_resample) or straight from the
// registers of weights_body (k_weights_pdf).
// what pdf_body reads from memory, requested in ONE go by pdf_load (k_weights_pdf: before get_weights, whose stores the compiler will not move
// loads across; written where they are used, the bins' copy into LDS alone was five round trips in a row, the ray's near / far / jitter and
// the first u another one behind the CDF)
struct PdfLoads {
  float bp[TN_MAX_SAMPLES / 64 + 1];  // the previous level's s-space bins, i = lane + 64 k
  float u0[TN_MAX_SAMPLES / 64 + 1];  // u_lin[j], j = lane + 64 k
  float near, far, jit;
};
__device__ __forceinline__ void pdf_load(const float* __restrict__ s_bins_prev, int Sp, const float* __restrict__ u_lin,
                                         const float* __restrict__ jitter, const float* __restrict__ nears, const float* __restrict__ fars, int S,
                                         int64_t ray, int lane, PdfLoads& L) {
  const float* bp = s_bins_prev + ray * (Sp + 1);
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) {
    const int i = lane + 64 * k;
    L.bp[k] = (i <= Sp) ? bp[i] : 0.0f;
    L.u0[k] = (i <= S) ? u_lin[i] : 0.0f;
  }
  L.near = nears[ray];
  L.far = fars[ray];
  L.jit = jitter != nullptr ? jitter[ray] : 0.0f;
}
template <int ITEMS>
__device__ __forceinline__ void pdf_body(const float (&w_raw)[ITEMS], const PdfLoads& L, int Sp, float anneal, bool jittered, int S, int64_t ray,
                                         float* __restrict__ s_bins, float* __restrict__ e_bins, float* cdf, float* pb, int lane,
                                         float* e_keep = nullptr, PdfLoads* next = nullptr) {
  float w[ITEMS];
  double loc = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    w[k] = 0.0f;
    if (i < Sp) {
      float x = w_raw[k];
      if (anneal != 1.0f) x = powf(x, anneal);  // torch.pow(weights, anneal); pow(x,1) is the identity
      w[k] = x + 0.01f;                        // histogram_padding
    }
    loc += (double)w[k];
  }
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k)
    if (lane + 64 * k <= Sp) pb[lane + 64 * k] = L.bp[k];
  float w_sum = (float)tn_wave_sum_d(loc);
  float padding = fmaxf(1e-5f - w_sum, 0.0f);
  float pad_each = padding / (float)Sp;
  w_sum = w_sum + padding;
  double ploc = 0.0;
  float pdf[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    pdf[k] = (i < Sp) ? (w[k] + pad_each) / w_sum : 0.0f;
    ploc += (double)pdf[k];
  }
  double incl = tn_wave_incl_scan_d(ploc, lane);
  double run = tn_excl_from_incl_d(incl, lane);
  if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i = lane * ITEMS + k;
    run += (double)pdf[k];
    if (i < Sp) cdf[i + 1] = fminf(1.0f, (float)run);
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float s_near = tn_spacing(L.near), s_far = tn_spacing(L.far);
  int nb = S + 1;
#pragma unroll
  for (int kk = 0; kk <= TN_MAX_SAMPLES / 64; ++kk) {
    const int j = lane + 64 * kk;
    if (j >= nb) continue;
    float u;
    if (jittered) u = L.u0[kk] + L.jit / (float)nb;
    else u = L.u0[kk] + (float)(1.0 / (2.0 * (double)nb));
    // searchsorted(cdf, u, side="right"): number of cdf entries <= u
    int lo = 0, hi = Sp + 1;
    while (lo < hi) {
      int m = (lo + hi) >> 1;
      if (cdf[m] <= u) lo = m + 1; else hi = m;
    }
    int below = lo - 1; below = below < 0 ? 0 : (below > Sp ? Sp : below);
    int above = lo > Sp ? Sp : lo;
    float c0 = cdf[below], c1 = cdf[above], b0 = pb[below], b1 = pb[above];
    float t = (u - c0) / (c1 - c0);
    t = tn_nan_to_num(t);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    float b = b0 + t * (b1 - b0);
    const float e = tn_s_to_euclid(b, s_near, s_far);
    s_bins[ray * nb + j] = b;
    e_bins[ray * nb + j] = e;
    // (the next-iteration sampling keeps the new bins for the stages that follow in the same wave: e in LDS, s in the registers the next
    // PDF stage reads them from -- bin j = lane + 64 kk is exactly PdfLoads::bp's layout)
    if (e_keep != nullptr) e_keep[j] = e;
    if (next != nullptr) next->bp[kk] = b;
  }
}
