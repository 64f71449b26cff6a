// The NEXT iteration's ray batch and proposal sampling as co-work of the optimiser launch (TnTrainStep::next_sampling).
//
// What tn_render_rays_train runs in front of the field -- CameraOptimizer.apply_to_raybundle (cameras/camera_optimizers.py:130-176), the level-0
// bins (model_components/ray_samplers.py:78-128,225-248), and twice density_fn -> get_weights -> PDFSampler (ray_samplers.py:577-618,276-372;
// fields/density_fields.py:95-118; cameras/rays.py:128-150) -- is a chain of five short launches that are bound by instruction issue and latency,
// not by memory: 86 us at the head of every iteration, beside an Adam pass over the field that is bound by HBM and reads none of their inputs.
// Per RAY the chain has no dependence on any other ray, so one wave can take a ray through ALL of it without a grid-wide step in between:
//   pixel + ground truth + ray (tn_sample_rays) -> pose correction -> 257 bins -> 256 proposal densities -> weights + PDF -> 97 bins ->
//   96 densities -> weights + PDF -> 49 bins
// with the bins and densities handed from stage to stage in the wave's slice of LDS / in registers, and every tensor the rest of the iteration
// reads (bins, densities, weights, median depths, the proposal encodings on update iterations) written to the next iteration's forward buffer
// exactly where tn_render_rays_train would have put it.  Every stage is the device function the stand-alone kernels call (tn_sampler_ray.h,
// tn_prop_point.h, tn_common.h, sample_ray_quad / pose_apply_ray above): the buffer is bit-identical to the in-line path's.
//
// What the chain reads of the model -- the proposal networks and the pose corrections -- must be final: tn_train_step steps those optimiser
// groups in a launch of their own in front of this one (a few MB), and the launch that carries the chain steps the rest (the field: 470 MB).
#pragma once
#include "tn_prop_point.h"
#include "tn_sampler_ray.h"

#define TN_WAVE_SYNC()                                     \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

// per wave: e bins [257] | densities [256] | cdf [257] | previous s bins [257], each region padded to 260 floats
#define NS_REGION 260
#define NS_WAVE_FLOATS (4 * NS_REGION)
#define NS_WEIGHT_FLOATS (PH * PROP_WROW + 8)  // one network's staged weights (prop_stage_weights), 16-byte multiple
#define NS_LDS_FLOATS (2 * NS_WEIGHT_FLOATS + 4 * NS_WAVE_FLOATS)

struct NextSamplingArgs {
  PropK p0, p1;
  const float* pose; const uint8_t* frozen; int num_cameras;
  const float* nears; const float* fars;
  const float *jit0, *jit1, *jit2;
  const float *lin0, *lin1, *lin2;
  float anneal;
  int S0, S1, S2;  // 128 < S0 <= 256 and 64 < S1 <= 128 (the lane layouts of tn_weights_resample for the default sampler); S2 <= 256
  int64_t N;
  float *origins, *directions;                 // pose-corrected rays
  float *s0, *e0, *d0, *w0, *m0;               // level 0: s / e bins [N,S0+1], density, weights [N,S0], median depth [N]
  float *s1, *e1, *d1, *w1, *m1;               // level 1
  float *s2, *e2;                              // the field's bins [N,S2+1]
  float *penc0, *penc1;                        // NULL, or the proposal levels' encodings (level-major [5][N*S] float2)
  int blocks;                                  // co-work blocks (4 rays each per trip); 0 = no chain
};

// enc_out: NULL (wave-uniform) = the encodings are not kept
__device__ __forceinline__ void ns_prop_level(const PropK& net, const float* s_w, const float (&o)[3], const float (&d)[3], const float* e_l, float* dn_l,
                                              int S, int64_t ray, int64_t N, float* __restrict__ density, float* __restrict__ enc_out, int lane) {
  const int64_t P = N * (int64_t)S;
#pragma unroll 1
  for (int i = lane; i < S; i += 64) {  // consecutive lanes = consecutive samples of the ray, as in k_prop_fwd
    const Contracted ct = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], e_l[i], e_l[i + 1]);
    const float dens = prop_density_sample<true, true>(net, s_w, ct.px, ct.py, ct.pz, ct.sel, ray * S + i, P, enc_out);
    density[ray * S + i] = dens;
    dn_l[i] = dens;
  }
}

template <int ITEMS>
__device__ __forceinline__ void ns_load_bins(const float* e_l, const float* dn_l, int S, int lane, float (&st)[ITEMS], float (&en)[ITEMS], float (&dn)[ITEMS]) {
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {  // ray_load_bins from the wave's LDS copies
    const int i = lane * ITEMS + k;
    st[k] = en[k] = dn[k] = 0.0f;
    if (i < S) { st[k] = e_l[i]; en[k] = e_l[i + 1]; dn[k] = dn_l[i]; }
  }
}

// one ray by one wave; lw: the wave's NS_WAVE_FLOATS floats of LDS; s_w0 / s_w1: the block's staged proposal weights
__device__ __forceinline__ void next_sampling_ray(const NextSamplingArgs& c, const SamplePixelsArgs& a, const RaygenArgs& g, int64_t ray, int lane, float* lw,
                                                  const float* s_w0, const float* s_w1) {
  float* e_l = lw;
  float* dn_l = lw + NS_REGION;
  float* cdf = lw + 2 * NS_REGION;
  float* pb = lw + 3 * NS_REGION;
  // ---- datamanager.next_train: pixel, ground truth, ray.  Every quad of the wave computes the same ray (quad lane q = undistortion of coordinate q);
  //      lane 0 stores.
  const SampledRay sr = sample_ray_quad(a, g, ray, lane, lane == 0);
  float din[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) din[q] = __shfl(sr.d[q], lane & ~3, 64);
  // ---- CameraOptimizer.apply_to_raybundle
  int64_t cam = (int64_t)__builtin_amdgcn_readfirstlane((int)sr.cam);  // (every lane holds the same camera; indices fit 32 bits)
  if (cam < 0 || cam >= c.num_cameras) cam = 0;
  float prow[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) prow[q] = c.pose[cam * 6 + q];
  float o[3], d[3];
  pose_apply_ray(prow, c.frozen != nullptr && c.frozen[cam], sr.o[0], sr.o[1], sr.o[2], din[0], din[1], din[2], o, d);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) { c.origins[ray * 3 + q] = o[q]; c.directions[ray * 3 + q] = d[q]; }
  }
  // ---- level 0: spaced bins
  const float near = c.nears[ray], far = c.fars[ray];
  const float s_near = tn_spacing(near), s_far = tn_spacing(far);
  PdfLoads L;
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) { L.bp[k] = 0.0f; L.u0[k] = (lane + 64 * k <= c.S1) ? c.lin1[lane + 64 * k] : 0.0f; }
  L.near = near; L.far = far; L.jit = c.jit1 != nullptr ? c.jit1[ray] : 0.0f;
  tn_spaced_bins_ray(c.lin0, c.jit0 != nullptr, c.jit0 != nullptr ? c.jit0[ray] : 0.0f, s_near, s_far, c.S0, c.s0 + ray * (c.S0 + 1), c.e0 + ray * (c.S0 + 1),
                     lane, e_l, L.bp);
  TN_WAVE_SYNC();
  ns_prop_level(c.p0, s_w0, o, d, e_l, dn_l, c.S0, ray, c.N, c.d0, c.penc0, lane);
  TN_WAVE_SYNC();
  PdfLoads L1;
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) { L1.bp[k] = 0.0f; L1.u0[k] = (lane + 64 * k <= c.S2) ? c.lin2[lane + 64 * k] : 0.0f; }
  L1.near = near; L1.far = far; L1.jit = c.jit2 != nullptr ? c.jit2[ray] : 0.0f;
  {  // get_weights of level 0 + PDF resampling -> level 1 (tn_weights_resample with S_prev in (128, 256]: 4 samples per lane)
    float st[4], en[4], dn[4], w[4];
    ns_load_bins<4>(e_l, dn_l, c.S0, lane, st, en, dn);
    weights_compute<4>(st, en, dn, c.S0, ray, c.w0, c.m0, lane, w);
    pdf_body<4>(w, L, c.S0, c.anneal, c.jit1 != nullptr, c.S1, ray, c.s1, c.e1, cdf, pb, lane, e_l, &L1);
  }
  TN_WAVE_SYNC();
  ns_prop_level(c.p1, s_w1, o, d, e_l, dn_l, c.S1, ray, c.N, c.d1, c.penc1, lane);
  TN_WAVE_SYNC();
  {  // level 1 -> the field's bins (S_prev in (64, 128]: 2 samples per lane)
    float st[2], en[2], dn[2], w[2];
    ns_load_bins<2>(e_l, dn_l, c.S1, lane, st, en, dn);
    weights_compute<2>(st, en, dn, c.S1, ray, c.w1, c.m1, lane, w);
    pdf_body<2>(w, L1, c.S1, c.anneal, c.jit2 != nullptr, c.S2, ray, c.s2, c.e2, cdf, pb, lane);
  }
  TN_WAVE_SYNC();  // (the next trip of this wave writes the regions again)
}

// the co-work row's block `bid` of `nblk` (256 threads = 4 rays per trip); lds: NS_LDS_FLOATS floats, 16-byte aligned
__device__ __forceinline__ void next_sampling_body(const NextSamplingArgs& c, const SamplePixelsArgs& a, const RaygenArgs& g, unsigned bid, unsigned nblk, float* lds) {
  float* s_w0 = lds;
  float* s_w1 = lds + NS_WEIGHT_FLOATS;
  for (int t = threadIdx.x; t < PH * PROP_WROW; t += blockDim.x) {
    const int j = t / PROP_WROW, k = t - j * PROP_WROW;
    s_w0[t] = k < PF ? c.p0.w0[j * PF + k] : (k == PF ? c.p0.b0[j] : c.p0.w1[j]);
    s_w1[t] = k < PF ? c.p1.w0[j * PF + k] : (k == PF ? c.p1.b0[j] : c.p1.w1[j]);
  }
  if (threadIdx.x == 0) { s_w0[PH * PROP_WROW] = c.p0.b1[0]; s_w1[PH * PROP_WROW] = c.p1.b1[0]; }
  __syncthreads();
  // (wave-uniform by construction; said so, the ray index and everything addressed by it live in scalar registers)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
  float* lw = lds + 2 * NS_WEIGHT_FLOATS + wv * NS_WAVE_FLOATS;
  for (int64_t ray = (int64_t)bid * wpb + wv; ray < c.N; ray += (int64_t)nblk * wpb) next_sampling_ray(c, a, g, ray, lane, lw, s_w0, s_w1);
}
