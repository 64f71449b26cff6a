// The proposal network's per-sample forward (HashMLPDensityField.get_density, fields/density_fields.py:95-118) as device functions shared by
// k_prop_fwd (tn_prop.hip) and the next-iteration sampling that rides in the optimiser launch (tn_next_sampling.h): one body, so the two paths
// produce the same bits.
#pragma once
#include "tn_common.h"

#define PL 5     // levels
#define PF 10    // PL * 2 features
#define PH 16    // hidden width
#define PROP_NW (PH * PF + PH + PH + 1)  // 193 weights: w0[16][10], b0[16], w1[16], b1

struct PropK {
  GridK g;
  const float *w0, *b0, *w1, *b1;
  float *gw0, *gb0, *gw1, *gb1;
};

// The 193 weights staged in LDS once per block, hidden units in PAIRS: pair q = units 2q, 2q + 1 takes 24 floats
//   [w0[2q][0] w0[2q+1][0] | w0[2q][1] w0[2q+1][1] | ... | w0[2q][9] w0[2q+1][9] | b0[2q] b0[2q+1] | w1[2q] w1[2q+1]],  then b1,
// read as six broadcast ds_read_b128 per pair (all lanes one address: no bank conflicts).  Adjacent floats = the same input weight of two hidden
// units = one operand of a packed FMA (v_pk_fma_f32: both units' accumulators advance per instruction; per component the same fmaf chain in the
// same order as the scalar form -- bit-identical).  Read through the kernel's pointers the weights end up as ~200 scalar registers the compiler
// spills into VGPR lanes: 698 v_readlane per sample made the kernel VALU-bound (45 us at level 0).
#define PROP_WROW 12
__device__ __forceinline__ int prop_weight_slot(int j, int k) { return (j >> 1) * (2 * PROP_WROW) + 2 * k + (j & 1); }  // k: 0..9 w0[j][k], 10 b0[j], 11 w1[j]
__device__ __forceinline__ void prop_stage_weights_into(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, float* s_w) {  // s_w: PH * PROP_WROW + 4 floats, 16-B aligned
  for (int t = threadIdx.x; t < PH * PROP_WROW; t += blockDim.x) {
    const int j = t / PROP_WROW, k = t - j * PROP_WROW;
    s_w[prop_weight_slot(j, k)] = k < PF ? w0[j * PF + k] : (k == PF ? b0[j] : w1[j]);
  }
  if (threadIdx.x == 0) s_w[PH * PROP_WROW] = b1[0];
}
__device__ __forceinline__ void prop_stage_weights(const PropK& net, float* s_w) {
  prop_stage_weights_into(net.w0, net.b0, net.w1, net.b1, s_w);
  __syncthreads();
}
// Linear(10,16) ReLU Linear(16,1) on the staged weights: out = b1 + sum_j w1[j] relu(b0[j] + sum_k w0[j][k] enc[k]), j ascending
__device__ __forceinline__ float prop_mlp(const float* s_w, const float (&enc)[PF]) {
  float out = s_w[PH * PROP_WROW];
#pragma unroll 2
  for (int q = 0; q < PH / 2; ++q) {
    const float4* row = reinterpret_cast<const float4*>(s_w + q * 2 * PROP_WROW);
    const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3], r4 = row[4], r5 = row[5];
    tn_v2f a = {r5.x, r5.y};  // b0 of the two units
    a = __builtin_elementwise_fma(tn_v2f{r0.x, r0.y}, tn_v2f{enc[0], enc[0]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r0.z, r0.w}, tn_v2f{enc[1], enc[1]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r1.x, r1.y}, tn_v2f{enc[2], enc[2]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r1.z, r1.w}, tn_v2f{enc[3], enc[3]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r2.x, r2.y}, tn_v2f{enc[4], enc[4]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r2.z, r2.w}, tn_v2f{enc[5], enc[5]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r3.x, r3.y}, tn_v2f{enc[6], enc[6]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r3.z, r3.w}, tn_v2f{enc[7], enc[7]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r4.x, r4.y}, tn_v2f{enc[8], enc[8]}, a);
    a = __builtin_elementwise_fma(tn_v2f{r4.z, r4.w}, tn_v2f{enc[9], enc[9]}, a);
    out = fmaf(r5.z, fmaxf(a.x, 0.0f), out);
    out = fmaf(r5.w, fmaxf(a.y, 0.0f), out);
  }
  return out;
}

// density of ONE sample: 5-level hash gather + trilinear -> Linear(10,16) ReLU Linear(16,1) -> trunc_exp -> * selector.  s_w: the staged weight
// rows of prop_stage_weights.  enc_out (SAVE_ENC): level-major [PL][P] float2, entry i.
// LEVEL_FENCE: nothing is scheduled across the boundary between two levels (8 gathers in flight, not 40) -- measured: no fewer registers, more latency; unused.
// OPTIONAL_ENC: enc_out may be NULL at run time (wave-uniform): one instantiation for both kinds of iteration.
// hook: called between the gathers and the MLP, once the encoding has arrived (tn_next_sampling.h requests a batch of the optimiser's data there:
// in flight while the MLP runs from registers and LDS, landed before the next sample's table loads are waited for)
struct PropNoHook { __device__ __forceinline__ void operator()() const {} };
template <bool SAVE_ENC, bool LEVEL_FENCE = false, bool OPTIONAL_ENC = false, class Hook = PropNoHook>
__device__ __forceinline__ float prop_density_sample(const PropK& net, const float* s_w, float px, float py, float pz, bool sel, int64_t i, int64_t P, float* enc_out,
                                                     Hook hook = Hook()) {
  float enc[PF];
#pragma unroll
  for (int l = 0; l < PL; ++l) {
    if (LEVEL_FENCE && l > 0) __builtin_amdgcn_sched_barrier(0);
    float2 v = tn_encode_level(net.g.table, px, py, pz, net.g.res[l], net.g.mask, (uint32_t)l * net.g.tsize);
    enc[2 * l] = v.x;
    enc[2 * l + 1] = v.y;
    // level-major [PL][P] float2: 512 contiguous bytes per wave and level.  (With a hook the stores follow it: the hook waits for the wave's
    // outstanding memory operations, and five fresh stores would be among them)
    if (__is_same(Hook, PropNoHook) && SAVE_ENC && (!OPTIONAL_ENC || enc_out != nullptr)) *reinterpret_cast<float2*>(enc_out + ((int64_t)l * P + i) * 2) = v;
  }
  if (!__is_same(Hook, PropNoHook)) {
    asm volatile("" :: "v"(enc[0]), "v"(enc[1]), "v"(enc[2]), "v"(enc[3]), "v"(enc[4]), "v"(enc[5]), "v"(enc[6]), "v"(enc[7]), "v"(enc[8]), "v"(enc[9]));
    hook();
    if (SAVE_ENC && (!OPTIONAL_ENC || enc_out != nullptr)) {
#pragma unroll
      for (int l = 0; l < PL; ++l) *reinterpret_cast<float2*>(enc_out + ((int64_t)l * P + i) * 2) = make_float2(enc[2 * l], enc[2 * l + 1]);
    }
  }
  const float out = prop_mlp(s_w, enc);
  return sel ? expf(out) : 0.0f * expf(out);  // exp(x) * selector (0*inf = nan kept as torch would)
}
