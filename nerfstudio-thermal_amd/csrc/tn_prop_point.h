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

// The 193 weights as rows of 12 floats per hidden unit, [w0[j][0..9] | b0[j] | w1[j]], staged in LDS once per block and read as three
// broadcast ds_read_b128 per hidden unit (all lanes one address: no bank conflicts).  Read through the kernel's pointers they end up as
// ~200 scalar registers the compiler spills into VGPR lanes: 698 v_readlane per sample made the kernel VALU-bound (45 us at level 0).
#define PROP_WROW 12
__device__ __forceinline__ void prop_stage_weights(const PropK& net, float* s_w) {  // s_w: PH * PROP_WROW + 4 floats, 16-B aligned
  for (int t = threadIdx.x; t < PH * PROP_WROW; t += blockDim.x) {
    const int j = t / PROP_WROW, k = t - j * PROP_WROW;
    s_w[t] = k < PF ? net.w0[j * PF + k] : (k == PF ? net.b0[j] : net.w1[j]);
  }
  if (threadIdx.x == 0) s_w[PH * PROP_WROW] = net.b1[0];
  __syncthreads();
}

// density of ONE sample: 5-level hash gather + trilinear -> Linear(10,16) ReLU Linear(16,1) -> trunc_exp -> * selector.  s_w: the staged weight
// rows of prop_stage_weights.  enc_out (SAVE_ENC): level-major [PL][P] float2, entry i.
// LEVEL_FENCE: nothing is scheduled across the boundary between two levels (8 gathers in flight, not 40) -- measured: no fewer registers, more latency; unused.
// OPTIONAL_ENC: enc_out may be NULL at run time (wave-uniform): one instantiation for both kinds of iteration.
template <bool SAVE_ENC, bool LEVEL_FENCE = false, bool OPTIONAL_ENC = false>
__device__ __forceinline__ float prop_density_sample(const PropK& net, const float* s_w, float px, float py, float pz, bool sel, int64_t i, int64_t P, float* enc_out) {
  float enc[PF];
#pragma unroll
  for (int l = 0; l < PL; ++l) {
    if (LEVEL_FENCE && l > 0) __builtin_amdgcn_sched_barrier(0);
    float2 v = tn_encode_level(net.g.table, px, py, pz, net.g.res[l], net.g.mask, (uint32_t)l * net.g.tsize);
    enc[2 * l] = v.x;
    enc[2 * l + 1] = v.y;
    if (SAVE_ENC && (!OPTIONAL_ENC || enc_out != nullptr)) *reinterpret_cast<float2*>(enc_out + ((int64_t)l * P + i) * 2) = v;  // level-major [PL][P] float2: 512 contiguous bytes per wave and level
  }
  float out = s_w[PH * PROP_WROW];
#pragma unroll 4
  for (int j = 0; j < PH; ++j) {
    const float4 wa = *reinterpret_cast<const float4*>(s_w + j * PROP_WROW), wb = *reinterpret_cast<const float4*>(s_w + j * PROP_WROW + 4),
                 wc = *reinterpret_cast<const float4*>(s_w + j * PROP_WROW + 8);
    float a = wc.z;
    a = fmaf(wa.x, enc[0], a);
    a = fmaf(wa.y, enc[1], a);
    a = fmaf(wa.z, enc[2], a);
    a = fmaf(wa.w, enc[3], a);
    a = fmaf(wb.x, enc[4], a);
    a = fmaf(wb.y, enc[5], a);
    a = fmaf(wb.z, enc[6], a);
    a = fmaf(wb.w, enc[7], a);
    a = fmaf(wc.x, enc[8], a);
    a = fmaf(wc.y, enc[9], a);
    out = fmaf(wc.w, fmaxf(a, 0.0f), out);
  }
  return sel ? expf(out) : 0.0f * expf(out);  // exp(x) * selector (0*inf = nan kept as torch would)
}
