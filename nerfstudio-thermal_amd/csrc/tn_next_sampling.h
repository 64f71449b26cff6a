// The NEXT iteration's ray batch and proposal sampling as co-work of the optimiser launch (TnTrainStep::next_sampling).
//
// What tn_render_rays_train runs in front of the field -- CameraOptimizer.apply_to_raybundle (cameras/camera_optimizers.py:130-176), the level-0
// bins (model_components/ray_samplers.py:78-128,225-248), and twice density_fn -> get_weights -> PDFSampler (ray_samplers.py:577-618,276-372;
// fields/density_fields.py:95-118; cameras/rays.py:128-150) -- is a chain of five short launches that are bound by instruction issue and latency,
// not by memory: 86 us at the head of every iteration, beside an Adam pass over the field that is bound by HBM and reads none of their inputs.
// Per RAY the chain has no dependence on any other ray, so one wave can take a ray through ALL of it without a grid-wide step in between:
//   pose correction -> 257 bins -> 256 proposal densities -> weights + PDF -> 97 bins -> 96 densities -> weights + PDF -> 49 bins
// with the bins and densities handed from stage to stage in the wave's slice of LDS / in registers, and every tensor the rest of the iteration
// reads (bins, densities, weights, median depths, the proposal encodings on update iterations) written to the next iteration's forward buffer
// exactly where tn_render_rays_train would have put it.  Every stage is the device function the stand-alone kernels call (tn_sampler_ray.h,
// tn_common.h, pose_apply_ray; tn_prop_point.h restates k_prop_fwd's sample): the buffer is bit-identical to the in-line path's
// (tests/test_datamanager_gpu.py).  The batch itself (tn_sample_rays) stays the 4-lanes-per-ray co-work it was, in the launch in front.
//
// The chain is bound by vector-instruction ISSUE (~12 k wave-instructions per ray, 4 rays per SIMD), which is why it can hide beside a launch
// that is bound by HBM -- and why everything that is the same for a whole ray is kept off the per-wave path: the pose's exponential map runs
// once per block (4 rays in 4 lanes), the arguments are few enough to stay in scalar registers.
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
#define NS_RAY_FLOATS 32  // the block's pose-corrected rays: [wave][o(3) . d(3) .]
#define NS_LDS_FLOATS (2 * NS_WEIGHT_FLOATS + NS_RAY_FLOATS + 4 * NS_WAVE_FLOATS)

// Everything the chain reads or writes, small enough to stay in scalar registers for the whole ray (the first version carried both networks' full
// kernel structs and sixteen 64-bit pointers: ~280 scalar values spilled into vector lanes, 1 400 lane moves per ray).
struct NsProp { const float2* table; uint32_t mask, tsize; float res[PL]; const float *w0, *b0, *w1, *b1; };
enum { NS_O = 0, NS_D, NS_S0, NS_E0, NS_D0, NS_W0, NS_M0, NS_S1, NS_E1, NS_D1, NS_W1, NS_M1, NS_S2, NS_E2, NS_PENC0, NS_PENC1, NS_SLOTS };
struct NextSamplingArgs {
  NsProp p0, p1;
  const float* pose; const uint8_t* frozen; int num_cameras;
  const float *rays_o, *rays_d; const int64_t* cam;  // the batch as tn_sample_rays left it (the first optimiser launch of the iteration carries that)
  const float* nears; const float* fars;
  const float *jit0, *jit1, *jit2;
  const float *lin0, *lin1, *lin2;
  float anneal;
  int S0, S1, S2;  // 128 < S0 <= 256 and 64 < S1 <= 128 (the lane layouts of tn_weights_resample for the default sampler); S2 <= 256
  int N;           // a multiple of 4 (every wave of a block makes the same number of trips)
  float* out;      // the NEXT iteration's forward buffer; off[]: float offsets of its regions (tn_render_rays_train_layout)
  uint32_t off[NS_SLOTS];
  int save_enc;    // keep the proposal levels' encodings (level-major [5][N*S] float2 at off[NS_PENC*])
  int blocks;      // co-work blocks (4 rays each per trip); 0 = no chain
};

// the proposal network of one level for the wave's ray; enc_out: NULL (wave-uniform) = the encodings are not kept
__device__ __forceinline__ void ns_prop_level(const NsProp& np, const float* s_w, const float (&o)[3], const float (&d)[3], const float* e_l, float* dn_l,
                                              int S, int ray, int N, float* __restrict__ density, float* __restrict__ enc_out, int lane) {
  PropK net;  // (what prop_density_sample reads: table, resolutions, mask, table size)
  net.g.table = np.table; net.g.mask = np.mask; net.g.tsize = np.tsize;
#pragma unroll
  for (int l = 0; l < PL; ++l) net.g.res[l] = np.res[l];
  const int64_t P = (int64_t)N * S;
#pragma unroll 1
  for (int i = lane; i < S; i += 64) {  // consecutive lanes = consecutive samples of the ray, as in k_prop_fwd
    const Contracted ct = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], e_l[i], e_l[i + 1]);
    const float dens = prop_density_sample<true, false, true>(net, s_w, ct.px, ct.py, ct.pz, ct.sel, (int64_t)ray * S + i, P, enc_out);
    density[(int64_t)ray * S + i] = dens;
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

// one ray by one wave; lw: the wave's NS_WAVE_FLOATS floats of LDS; s_w0 / s_w1: the block's staged proposal weights; o / d: the pose-corrected ray
__device__ __forceinline__ void next_sampling_ray(const NextSamplingArgs& c, int ray, int lane, float* lw, const float* s_w0, const float* s_w1,
                                                  const float (&o)[3], const float (&d)[3]) {
  float* e_l = lw;
  float* dn_l = lw + NS_REGION;
  float* cdf = lw + 2 * NS_REGION;
  float* pb = lw + 3 * NS_REGION;
  auto at = [&](int slot) { return c.out + c.off[slot]; };
  // ---- level 0: spaced bins
  const float near = c.nears[ray], far = c.fars[ray];
  const float s_near = tn_spacing(near), s_far = tn_spacing(far);
  PdfLoads L;
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) { L.bp[k] = 0.0f; L.u0[k] = (lane + 64 * k <= c.S1) ? c.lin1[lane + 64 * k] : 0.0f; }
  L.near = near; L.far = far; L.jit = c.jit1 != nullptr ? c.jit1[ray] : 0.0f;
  tn_spaced_bins_ray(c.lin0, c.jit0 != nullptr, c.jit0 != nullptr ? c.jit0[ray] : 0.0f, s_near, s_far, c.S0, at(NS_S0) + (int64_t)ray * (c.S0 + 1),
                     at(NS_E0) + (int64_t)ray * (c.S0 + 1), lane, e_l, L.bp);
  TN_WAVE_SYNC();
  ns_prop_level(c.p0, s_w0, o, d, e_l, dn_l, c.S0, ray, c.N, at(NS_D0), c.save_enc ? at(NS_PENC0) : nullptr, lane);
  TN_WAVE_SYNC();
  PdfLoads L1;
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) { L1.bp[k] = 0.0f; L1.u0[k] = (lane + 64 * k <= c.S2) ? c.lin2[lane + 64 * k] : 0.0f; }
  L1.near = near; L1.far = far; L1.jit = c.jit2 != nullptr ? c.jit2[ray] : 0.0f;
  {  // get_weights of level 0 + PDF resampling -> level 1 (tn_weights_resample with S_prev in (128, 256]: 4 samples per lane)
    float st[4], en[4], dn[4], w[4];
    ns_load_bins<4>(e_l, dn_l, c.S0, lane, st, en, dn);
    weights_compute<4>(st, en, dn, c.S0, (int64_t)ray, at(NS_W0), at(NS_M0), lane, w);
    pdf_body<4>(w, L, c.S0, c.anneal, c.jit1 != nullptr, c.S1, (int64_t)ray, at(NS_S1), at(NS_E1), cdf, pb, lane, e_l, &L1);
  }
  TN_WAVE_SYNC();
  ns_prop_level(c.p1, s_w1, o, d, e_l, dn_l, c.S1, ray, c.N, at(NS_D1), c.save_enc ? at(NS_PENC1) : nullptr, lane);
  TN_WAVE_SYNC();
  {  // level 1 -> the field's bins (S_prev in (64, 128]: 2 samples per lane)
    float st[2], en[2], dn[2], w[2];
    ns_load_bins<2>(e_l, dn_l, c.S1, lane, st, en, dn);
    weights_compute<2>(st, en, dn, c.S1, (int64_t)ray, at(NS_W1), at(NS_M1), lane, w);
    pdf_body<2>(w, L1, c.S1, c.anneal, c.jit2 != nullptr, c.S2, (int64_t)ray, at(NS_S2), at(NS_E2), cdf, pb, lane);
  }
  TN_WAVE_SYNC();  // (the next trip of this wave writes the regions again)
}

// the co-work row's block `bid` of `nblk` (256 threads = 4 rays per trip); lds: NS_LDS_FLOATS floats, 16-byte aligned
__device__ __forceinline__ void next_sampling_body(const NextSamplingArgs& c, unsigned bid, unsigned nblk, float* lds) {
  float* s_w0 = lds;
  float* s_w1 = lds + NS_WEIGHT_FLOATS;
  float* s_ray = lds + 2 * NS_WEIGHT_FLOATS;
  prop_stage_weights_into(c.p0.w0, c.p0.b0, c.p0.w1, c.p0.b1, s_w0);
  prop_stage_weights_into(c.p1.w0, c.p1.b0, c.p1.w1, c.p1.b1, s_w1);
  // (wave-uniform by construction; said so, the ray index and everything addressed by it live in scalar registers)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
  float* lw = lds + 2 * NS_WEIGHT_FLOATS + NS_RAY_FLOATS + wv * NS_WAVE_FLOATS;
  for (int r0 = (int)bid * wpb; r0 < c.N; r0 += (int)nblk * wpb) {
    // ---- CameraOptimizer.apply_to_raybundle of the block's rays by its first lanes (exp_map_SO3xR3 is ~500 instructions: once per block, not once
    //      per wave; every wave of the block makes the same trips -- N is a multiple of the rays per block -- so the barriers match)
    if (threadIdx.x < (unsigned)wpb) {
      const int r = r0 + (int)threadIdx.x;
      int64_t cam = c.cam[r];
      if (cam < 0 || cam >= c.num_cameras) cam = 0;
      float prow[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) prow[q] = c.pose[cam * 6 + q];
      float o[3], d[3];
      pose_apply_ray(prow, c.frozen != nullptr && c.frozen[cam], c.rays_o[r * 3], c.rays_o[r * 3 + 1], c.rays_o[r * 3 + 2], c.rays_d[r * 3], c.rays_d[r * 3 + 1],
                     c.rays_d[r * 3 + 2], o, d);
      float* po = c.out + c.off[NS_O] + (int64_t)r * 3;
      float* pd = c.out + c.off[NS_D] + (int64_t)r * 3;
#pragma unroll
      for (int q = 0; q < 3; ++q) { po[q] = o[q]; pd[q] = d[q]; s_ray[threadIdx.x * 8 + q] = o[q]; s_ray[threadIdx.x * 8 + 4 + q] = d[q]; }
    }
    __syncthreads();  // (first trip: also the staged weights)
    float o[3], d[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { o[q] = s_ray[wv * 8 + q]; d[q] = s_ray[wv * 8 + 4 + q]; }
    next_sampling_ray(c, r0 + wv, lane, lw, s_w0, s_w1, o, d);
    __syncthreads();  // (the next trip's rays overwrite s_ray)
  }
}
