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

// ---- Adam inside the chain's waves (TN_NEXT_SAMPLING=4, an experiment: profiles/r06_next_sampling.md) -------------------------------------------
// The launch is a SUM because the chain's 1024 blocks fill every wave slot of the chip until they retire.  Here the chain's waves step a part of
// the field's optimiser range themselves, between their own stages: a wave requests one batch of 64 float4 of g, m, v and p with
// global_load_lds_dwordx4 -- straight into a private 4-KB slot of LDS, no registers held while the loads fly (the chain has none to spare) --
// goes on with its ray, and at its next service point reads the batch back, runs torch.optim.Adam's arithmetic (adam_range_body's expressions,
// term for term) and stores.  Batches are dealt round-robin: batch q * W + (wave index) at the wave's q-th service point (W = waves of the row);
// what a wave has not reached at the end of its rays it steps there, one batch at a time.  The launch's range row takes the rest of the range.
struct FusedAdam {
  float* p; const float* g; float* m; float* v;  // the range's first element in the four arenas
  uint32_t batches;                              // 256-float batches [0, batches) stepped by the chain's waves (0: none)
  int32_t flag, step, max_steps, sched_step, lag_index, zero_g;
  double beta1, beta2, lr, lr_final;
  float eps;
  const float* found_inf; const int32_t* skipped;
};
#define FA_STAGE_FLOATS 1024  // per wave: g | m | v | p, 64 lanes x float4 each
#define FA_NONE 0xffffffffu
#define FA_SITES_PER_RAY 10   // batches a ray's stages take: 4 + 2 proposal trips, 2 x (get_weights, PDF) of next_sampling_ray (the host sizes `batches` by it)
struct FaCtx {
  const FusedAdam* fa; float* stage; uint32_t W, next, pend; float ns, bc; int skip;
};
typedef float fa_v4f __attribute__((ext_vector_type(4)));
// The four requests of a batch, written in assembly: through the builtin the compiler KNOWS that LDS is being written behind its back and, unable to
// tell the staging slots from the chain's own LDS regions, waits for the loads (vmcnt(0)) in front of the chain's very next LDS read -- the request
// then hides behind a few dozen instructions, not behind a stage.  What the compiler does not see it does not wait for; fa_process waits itself.
// (Its own vmcnt bookkeeping for the chain's loads stays safe: four unseen older requests only make a vmcnt(N) wait longer, never shorter.)
// Addresses: wave-uniform 64-bit bases in scalar registers + ONE 32-bit byte offset per lane (a batch lies within 4 GB of the range's start).
__device__ __forceinline__ void fa_issue(const FusedAdam& fa, float* stage, uint32_t vb, int lane) {
  const uint32_t boff = vb * 1024u + (uint32_t)lane * 16u;
  const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)stage;  // (wave-uniform LDS byte address)
  uint32_t saved;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"  // (the previous batch's reads of the slot are done)
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %6 nt\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %7 nt\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %8 nt\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %9\n\t"  // (the parameters: cached like adam_range_body's plain load -- the next forward gathers from them)
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "s"(l0), "s"(l0 + 1024u), "s"(l0 + 2048u), "s"(l0 + 3072u), "v"(boff), "s"(fa.g), "s"(fa.m), "s"(fa.v), "s"(fa.p)
      : "memory");
}
// The batch from LDS, stepped and stored in two halves of 512 bytes (lane l: floats 2 l, 2 l + 1 of the half -- 8 contiguous bytes per lane and
// access): half the registers of a float4 per lane, and the chain has none to spare (a spilled register's reload is a vector memory operation that
// waits for everything before it -- the requests in flight included).
typedef float fa_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fa_process(const FusedAdam& fa, float* stage, uint32_t vb, int lane, float neg_step, float bc2_sqrt, int skip) {
  const uint32_t boff = vb * 1024u + (uint32_t)lane * 8u;  // byte offset of the lane's pair in the first half
  char* gb = reinterpret_cast<char*>(const_cast<float*>(fa.g));
  // (the zeros are made where they are stored: as a constant the compiler materialises one register pair per service point at the top of the
  // kernel and keeps -- spills -- them all)
#define FA_ZERO2(Z) fa_v2f Z; asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(Z.x), "=v"(Z.y))
  if (skip) {  // GradScaler found an inf in the group: nothing is stepped, the gradients are consumed
    if (fa.zero_g) {
      FA_ZERO2(z);
      __builtin_nontemporal_store(z, reinterpret_cast<fa_v2f*>(gb + boff));
      __builtin_nontemporal_store(z, reinterpret_cast<fa_v2f*>(gb + boff + 512u));
    }
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the batch is in LDS
  const float b1 = (float)fa.beta1, b2 = (float)fa.beta2, omb1 = (float)(1.0 - fa.beta1), omb2 = (float)(1.0 - fa.beta2), eps = fa.eps;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const float* sl = stage + hf * 128 + lane * 2;
    const fa_v2f gg = *reinterpret_cast<const fa_v2f*>(sl);
    fa_v2f mm = *reinterpret_cast<const fa_v2f*>(sl + 256), vv = *reinterpret_cast<const fa_v2f*>(sl + 512);
    typedef uint32_t fa_v2u __attribute__((ext_vector_type(2)));
    const fa_v2u zb = __builtin_bit_cast(fa_v2u, gg) | __builtin_bit_cast(fa_v2u, mm) | __builtin_bit_cast(fa_v2u, vv);
    if ((zb.x | zb.y) != 0u) {  // (entries that never received a gradient stay as they are: adam_range_body)
      fa_v2f pp = *reinterpret_cast<const fa_v2f*>(sl + 768);
#define FA_ADAM1(C)                                     \
  {                                                     \
    const float g_ = gg.C;                              \
    const float m_new_ = mm.C * b1 + g_ * omb1;         \
    const float v_new_ = vv.C * b2 + (omb2 * g_) * g_;  \
    const float den_ = sqrtf(v_new_) / bc2_sqrt + eps;  \
    pp.C = pp.C + neg_step * (m_new_ / den_);           \
    mm.C = m_new_;                                      \
    vv.C = v_new_;                                      \
  }
      FA_ADAM1(x) FA_ADAM1(y)
#undef FA_ADAM1
      const uint32_t o = boff + hf * 512u;
      *reinterpret_cast<fa_v2f*>(reinterpret_cast<char*>(fa.p) + o) = pp;
      __builtin_nontemporal_store(mm, reinterpret_cast<fa_v2f*>(reinterpret_cast<char*>(fa.m) + o));
      __builtin_nontemporal_store(vv, reinterpret_cast<fa_v2f*>(reinterpret_cast<char*>(fa.v) + o));
      if (fa.zero_g) { FA_ZERO2(z); __builtin_nontemporal_store(z, reinterpret_cast<fa_v2f*>(gb + o)); }
    }
  }
#undef FA_ZERO2
}
// A batch is REQUESTED where a stretch of the chain begins that waits for no global load of its own (the proposal MLP behind a sample's gathers,
// get_weights, the PDF stage) and STEPPED at a later point where the chain has just waited for its own loads anyway (behind the next trip's
// gathers; behind the PDF inputs' loads): a request in flight is the oldest vector-memory operation of the wave, so any wait for a later load --
// or, in fa_process, for the chain's own fresh stores -- would wait for it too.  All of it is wave-uniform control.
__device__ __forceinline__ void fa_request(FaCtx& x, int lane) {
  if (x.pend == FA_NONE && x.next < x.fa->batches) {
    if (!x.skip) fa_issue(*x.fa, x.stage, x.next, lane);
    x.pend = x.next;
    x.next += x.W;
  }
}
__device__ __forceinline__ void fa_step(FaCtx& x, int lane) {
  if (x.pend != FA_NONE) fa_process(*x.fa, x.stage, x.pend, lane, x.ns, x.bc, x.skip);
  x.pend = FA_NONE;
}
struct FaHook {
  FaCtx* x; int lane;
  __device__ __forceinline__ void operator()() const { fa_step(*x, lane); fa_request(*x, lane); }
};

// the proposal network of one level for the wave's ray; enc_out: NULL (wave-uniform) = the encodings are not kept
template <bool FUSED = false>
__device__ __forceinline__ void ns_prop_level(const NsProp& np, const float* s_w, const float (&o)[3], const float (&d)[3], const float* e_l, float* dn_l,
                                              int S, int ray, int N, float* __restrict__ density, float* __restrict__ enc_out, int lane, FaCtx* fx = nullptr) {
  PropK net;  // (what prop_density_sample reads: table, resolutions, mask, table size)
  net.g.table = np.table; net.g.mask = np.mask; net.g.tsize = np.tsize;
#pragma unroll
  for (int l = 0; l < PL; ++l) net.g.res[l] = np.res[l];
  const int64_t P = (int64_t)N * S;
  if (FUSED) {
    // the same trips with EVERY lane in them (the service point inside needs the whole wave): lanes beyond the last sample redo the last sample
    // -- the same bits to the same addresses where the sample code stores -- and keep their result to themselves
#pragma unroll 1
    for (int base = 0; base < S; base += 64) {
      const bool act = base + lane < S;
      const int i = act ? base + lane : S - 1;
      const Contracted ct = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], e_l[i], e_l[i + 1]);
      const float dens = prop_density_sample<true, false, true, FaHook>(net, s_w, ct.px, ct.py, ct.pz, ct.sel, (int64_t)ray * S + i, P, enc_out, FaHook{fx, lane});
      if (act) { density[(int64_t)ray * S + i] = dens; dn_l[i] = dens; }
    }
    return;
  }
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
template <bool FUSED = false>
__device__ __forceinline__ void next_sampling_ray(const NextSamplingArgs& c, int ray, int lane, float* lw, const float* s_w0, const float* s_w1,
                                                  const float (&o)[3], const float (&d)[3], FaCtx* fx = nullptr) {
#define FA_REQUEST() do { if (FUSED) fa_request(*fx, lane); } while (0)
#define FA_STEP() do { if (FUSED) fa_step(*fx, lane); } while (0)
  float* e_l = lw;
  float* dn_l = lw + NS_REGION;
  float* cdf = lw + 2 * NS_REGION;
  float* pb = lw + 3 * NS_REGION;
  auto at = [&](int slot) { return c.out + c.off[slot]; };
  // ---- level 0: spaced bins
  const float near = c.nears[ray], far = c.fars[ray];
  const float s_near = tn_spacing(near), s_far = tn_spacing(far);
  PdfLoads L;
  // (FUSED: the PDF stages' inputs are requested where they are used, not a stage ahead -- 16 registers the optimiser's service points need; a
  // spilled register's reload would wait for the requests in flight)
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) { L.bp[k] = 0.0f; L.u0[k] = (!FUSED && lane + 64 * k <= c.S1) ? c.lin1[lane + 64 * k] : 0.0f; }
  L.near = near; L.far = far; L.jit = (!FUSED && c.jit1 != nullptr) ? c.jit1[ray] : 0.0f;
  tn_spaced_bins_ray(c.lin0, c.jit0 != nullptr, c.jit0 != nullptr ? c.jit0[ray] : 0.0f, s_near, s_far, c.S0, at(NS_S0) + (int64_t)ray * (c.S0 + 1),
                     at(NS_E0) + (int64_t)ray * (c.S0 + 1), lane, e_l, L.bp);
  TN_WAVE_SYNC();
  ns_prop_level<FUSED>(c.p0, s_w0, o, d, e_l, dn_l, c.S0, ray, c.N, at(NS_D0), c.save_enc ? at(NS_PENC0) : nullptr, lane, fx);
  TN_WAVE_SYNC();
  PdfLoads L1;
#pragma unroll
  for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) { L1.bp[k] = 0.0f; L1.u0[k] = (!FUSED && lane + 64 * k <= c.S2) ? c.lin2[lane + 64 * k] : 0.0f; }
  L1.near = near; L1.far = far; L1.jit = (!FUSED && c.jit2 != nullptr) ? c.jit2[ray] : 0.0f;
  if (FUSED) {
#pragma unroll
    for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) L.u0[k] = (lane + 64 * k <= c.S1) ? c.lin1[lane + 64 * k] : 0.0f;
    L.jit = c.jit1 != nullptr ? c.jit1[ray] : 0.0f;
    asm volatile("" :: "v"(L.u0[0]), "v"(L.u0[1]), "v"(L.u0[2]), "v"(L.u0[3]), "v"(L.u0[4]), "v"(L.jit));  // (arrived: nothing the stage waits for is behind the request)
  }
  FA_STEP();
  FA_REQUEST();
  {  // get_weights of level 0 + PDF resampling -> level 1 (tn_weights_resample with S_prev in (128, 256]: 4 samples per lane)
    float st[4], en[4], dn[4], w[4];
    ns_load_bins<4>(e_l, dn_l, c.S0, lane, st, en, dn);
    weights_compute<4>(st, en, dn, c.S0, (int64_t)ray, at(NS_W0), at(NS_M0), lane, w);
    FA_STEP();
    FA_REQUEST();
    pdf_body<4>(w, L, c.S0, c.anneal, c.jit1 != nullptr, c.S1, (int64_t)ray, at(NS_S1), at(NS_E1), cdf, pb, lane, e_l, &L1);
  }
  TN_WAVE_SYNC();
  ns_prop_level<FUSED>(c.p1, s_w1, o, d, e_l, dn_l, c.S1, ray, c.N, at(NS_D1), c.save_enc ? at(NS_PENC1) : nullptr, lane, fx);
  TN_WAVE_SYNC();
  if (FUSED) {
#pragma unroll
    for (int k = 0; k <= TN_MAX_SAMPLES / 64; ++k) L1.u0[k] = (lane + 64 * k <= c.S2) ? c.lin2[lane + 64 * k] : 0.0f;
    L1.jit = c.jit2 != nullptr ? c.jit2[ray] : 0.0f;
    asm volatile("" :: "v"(L1.u0[0]), "v"(L1.u0[1]), "v"(L1.u0[2]), "v"(L1.u0[3]), "v"(L1.u0[4]), "v"(L1.jit));
  }
  FA_STEP();
  FA_REQUEST();
  {  // level 1 -> the field's bins (S_prev in (64, 128]: 2 samples per lane)
    float st[2], en[2], dn[2], w[2];
    ns_load_bins<2>(e_l, dn_l, c.S1, lane, st, en, dn);
    weights_compute<2>(st, en, dn, c.S1, (int64_t)ray, at(NS_W1), at(NS_M1), lane, w);
    FA_STEP();
    FA_REQUEST();
    pdf_body<2>(w, L1, c.S1, c.anneal, c.jit2 != nullptr, c.S2, (int64_t)ray, at(NS_S2), at(NS_E2), cdf, pb, lane);
  }
  TN_WAVE_SYNC();  // (the next trip of this wave writes the regions again)
#undef FA_REQUEST
#undef FA_STEP
}

// the co-work row's block `bid` of `nblk` (256 threads = 4 rays per trip); lds: NS_LDS_FLOATS floats, 16-byte aligned
// fa / fa_stage (FUSED): the optimiser range the waves step on their way, and the block's [waves][FA_STAGE_FLOATS] staging slots (an LDS array of
// its own: the chain's reads of ITS regions then never wait for a batch in flight)
template <bool FUSED = false>
__device__ __forceinline__ void next_sampling_body(const NextSamplingArgs& c, unsigned bid, unsigned nblk, float* lds, const FusedAdam* fa = nullptr,
                                                   float* fa_stage = nullptr, float* fa_scal = nullptr) {
  float* s_w0 = lds;
  float* s_w1 = lds + NS_WEIGHT_FLOATS;
  float* s_ray = lds + 2 * NS_WEIGHT_FLOATS;
  prop_stage_weights_into(c.p0.w0, c.p0.b0, c.p0.w1, c.p0.b1, s_w0);
  prop_stage_weights_into(c.p1.w0, c.p1.b0, c.p1.w1, c.p1.b1, s_w1);
  // (wave-uniform by construction; said so, the ray index and everything addressed by it live in scalar registers)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
  float* lw = lds + 2 * NS_WEIGHT_FLOATS + NS_RAY_FLOATS + wv * NS_WAVE_FLOATS;
  FaCtx fx{};
  if (FUSED) {
    if (threadIdx.x == 0) {  // the step's scalars, as adam_range_body's thread 0 forms them
      const int fl = fa->flag;
      const int lag = (fa->skipped != nullptr && fa->lag_index >= 0) ? fa->skipped[fa->lag_index] : 0;
      const int sk = fa->skipped ? fa->skipped[fl] : 0;
      const int eff = fa->step - sk;
      const double bc1 = 1.0 - pow(fa->beta1, (double)(eff < 1 ? 1 : eff)), bc2 = 1.0 - pow(fa->beta2, (double)(eff < 1 ? 1 : eff));
      double lr = fa->lr;
      if (fa->max_steps > 0) {
        double t = (double)(fa->sched_step - lag) / (double)fa->max_steps;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        lr = exp(log(fa->lr) * (1.0 - t) + log(fa->lr_final) * t);
      }
      fa_scal[0] = (float)(-(lr / bc1));
      fa_scal[1] = (float)sqrt(bc2);
      fa_scal[2] = (fa->found_inf != nullptr && fa->found_inf[fl] != 0.0f) ? 1.0f : 0.0f;
    }
    fx.fa = fa;
    fx.stage = fa_stage + wv * FA_STAGE_FLOATS;
    fx.W = nblk * (unsigned)wpb;
    fx.next = bid * (unsigned)wpb + (unsigned)wv;
    fx.pend = FA_NONE;
  }
  bool first_trip = true;
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
    if (FUSED && first_trip) {
      fx.ns = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fa_scal[0])));
      fx.bc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fa_scal[1])));
      fx.skip = __builtin_amdgcn_readfirstlane(fa_scal[2] != 0.0f ? 1 : 0);
      first_trip = false;
    }
    float o[3], d[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { o[q] = s_ray[wv * 8 + q]; d[q] = s_ray[wv * 8 + 4 + q]; }
    next_sampling_ray<FUSED>(c, r0 + wv, lane, lw, s_w0, s_w1, o, d, &fx);
    __syncthreads();  // (the next trip's rays overwrite s_ray)
  }
  if (FUSED) {
    if (first_trip) {  // (a block without rays still has its share of the range)
      __syncthreads();
      fx.ns = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fa_scal[0])));
      fx.bc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fa_scal[1])));
      fx.skip = __builtin_amdgcn_readfirstlane(fa_scal[2] != 0.0f ? 1 : 0);
    }
    while (fx.pend != FA_NONE || fx.next < fa->batches) { fa_request(fx, lane); fa_step(fx, lane); }  // what the wave's rays did not reach, one batch at a time
  }
}
