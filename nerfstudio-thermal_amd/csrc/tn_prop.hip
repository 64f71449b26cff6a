// Proposal-network density: one lane per sample point, everything fused (position -> contraction -> selector ->
// 5-level hash gather + trilinear -> Linear(10,16) ReLU Linear(16,1) -> trunc_exp -> *selector).
// Replaces Field.density_fn -> HashMLPDensityField.get_density (fields/base_field.py:48-68, fields/density_fields.py:95-118).
// Consecutive lanes are consecutive samples of one ray, so neighbouring lanes hit the same / adjacent grid cells on the
// coarse levels (gathers coalesce in the texture-address unit) and the 772 B of MLP weights are wave-uniform scalar loads.
#include "tn_common.h"

#define PL 5     // levels
#define PF 10    // PL * 2 features
#define PH 16    // hidden width
#define PROP_NW (PH * PF + PH + PH + 1)  // 193 weights: w0[16][10], b0[16], w1[16], b1

int tn_wgrad_launch2(const float* dY0, int ldy0, int out0, const float* X0, int ldx0, int in0, float* dW0, int ldw0, float* db0, const float* dY1,
                     int ldy1, int out1, const float* X1, int ldx1, int in1, float* dW1, int ldw1, float* db1, int64_t P, hipStream_t stream);  // tn_field.hip

struct PropK {
  GridK g;
  const float *w0, *b0, *w1, *b1;
  float *gw0, *gb0, *gw1, *gb1;
};

__global__ void __launch_bounds__(256) k_prop_fwd(PropK net, const float* __restrict__ origins, const float* __restrict__ directions,
                                                  const float* __restrict__ e_bins, int64_t N, int S, float* __restrict__ density) {
  int64_t P = N * (int64_t)S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t ray = i / S;
    int s = (int)(i - ray * S);
    const float* o = origins + ray * 3;
    const float* d = directions + ray * 3;
    const float* eb = e_bins + ray * (S + 1) + s;
    Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], eb[0], eb[1]);
    float enc[PF];
#pragma unroll
    for (int l = 0; l < PL; ++l) {
      float2 v = tn_encode_level(net.g.table, c.px, c.py, c.pz, net.g.res[l], net.g.mask, (uint32_t)l * net.g.tsize);
      enc[2 * l] = v.x;
      enc[2 * l + 1] = v.y;
    }
    float out = net.b1[0];
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      float a = net.b0[j];
#pragma unroll
      for (int k = 0; k < PF; ++k) a = fmaf(net.w0[j * PF + k], enc[k], a);
      a = fmaxf(a, 0.0f);
      out = fmaf(net.w1[j], a, out);
    }
    density[i] = c.sel ? expf(out) : 0.0f * expf(out);  // exp(x) * selector (0*inf = nan kept as torch would)
  }
}

extern "C" int tn_prop_density_fwd(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins, int64_t N,
                                   int32_t S, float* density, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(net && origins && directions && e_bins && density, "tn_prop_density_fwd: null pointer");
  TN_REQUIRE(net->grid.table && net->w0 && net->b0 && net->w1 && net->b1, "tn_prop_density_fwd: null parameter pointer");
  TN_REQUIRE(net->grid.num_levels == PL, "tn_prop_density_fwd: proposal grids are built for %d levels, got %d", PL, net->grid.num_levels);
  TN_REQUIRE(net->grid.log2_hashmap_size >= 1 && net->grid.log2_hashmap_size <= 24, "tn_prop_density_fwd: bad log2_hashmap_size");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_prop_density_fwd: bad N=%lld S=%d", (long long)N, S);
  if (N == 0) return TN_OK;
  PropK k{make_gridk(net->grid), net->w0, net->b0, net->w1, net->b1, nullptr, nullptr, nullptr, nullptr};
  int64_t P = N * (int64_t)S;
  int grid = (int)std::min<int64_t>(tn_cdiv(P, 256), 256 * 16);
  hipLaunchKernelGGL(k_prop_fwd, dim3(grid), dim3(256), 0, tn_s(stream), k, origins, directions, e_bins, N, S, density);
  TN_CHECK_LAUNCH("tn_prop_density_fwd");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ backward
// k_prop_bwd_mlp (1 lane = 1 sample): recompute the forward, then  d_out = g * exp(clamp(out,-15,15)) * sel  (trunc_exp backward),
//   d a_j = d_out*w1_j*[a_j>0],  d enc_k = sum_j d a_j w0_jk.  It writes the operands of the two weight-gradient GEMMs
//   (dW0 = dA^T ENC, dW1 = dOUT^T H; K = all points; reduced by the shared fp32-MFMA kernel tn_wgrad_launch) and d enc.
// The table scatter-add (+ d position) is the shared request-coalescing kernel in tn_scatter.hip.
__global__ void __launch_bounds__(256) k_prop_bwd_mlp(PropK net, const float* __restrict__ origins, const float* __restrict__ directions,
                                                      const float* __restrict__ e_bins, const float* __restrict__ d_density, int64_t N, int S,
                                                      float* __restrict__ ws_da, float* __restrict__ ws_dout, float* __restrict__ ws_enc,
                                                      float* __restrict__ ws_h, float* __restrict__ ws_denc) {
  int64_t P = N * (int64_t)S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t ray = i / S;
    int s = (int)(i - ray * S);
    const float* o = origins + ray * 3;
    const float* d = directions + ray * 3;
    const float* eb = e_bins + ray * (S + 1) + s;
    Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], eb[0], eb[1]);
    float enc[PF];
#pragma unroll
    for (int l = 0; l < PL; ++l) {
      float2 v = tn_encode_level(net.g.table, c.px, c.py, c.pz, net.g.res[l], net.g.mask, (uint32_t)l * net.g.tsize);
      enc[2 * l] = v.x;
      enc[2 * l + 1] = v.y;
    }
    float a[PH];
    float out = net.b1[0];
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      float t = net.b0[j];
#pragma unroll
      for (int k = 0; k < PF; ++k) t = fmaf(net.w0[j * PF + k], enc[k], t);
      a[j] = t;
      out = fmaf(net.w1[j], fmaxf(t, 0.0f), out);
    }
    float d_out = c.sel ? d_density[i] * expf(fminf(fmaxf(out, -15.0f), 15.0f)) : 0.0f;
    float denc[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) denc[k] = 0.0f;
    float da[PH], hh[PH];
#pragma unroll
    for (int j = 0; j < PH; ++j) {
      hh[j] = fmaxf(a[j], 0.0f);
      da[j] = (a[j] > 0.0f) ? d_out * net.w1[j] : 0.0f;
#pragma unroll
      for (int k = 0; k < PF; ++k) denc[k] = fmaf(da[j], net.w0[j * PF + k], denc[k]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(ws_da + i * 16 + 4 * q) = make_float4(da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]);
      *reinterpret_cast<float4*>(ws_h + i * 16 + 4 * q) = make_float4(hh[4 * q], hh[4 * q + 1], hh[4 * q + 2], hh[4 * q + 3]);
    }
    *reinterpret_cast<float4*>(ws_enc + i * 16 + 0) = make_float4(enc[0], enc[1], enc[2], enc[3]);
    *reinterpret_cast<float4*>(ws_enc + i * 16 + 4) = make_float4(enc[4], enc[5], enc[6], enc[7]);
    *reinterpret_cast<float4*>(ws_enc + i * 16 + 8) = make_float4(enc[8], enc[9], 0.0f, 0.0f);
    *reinterpret_cast<float4*>(ws_denc + i * 16 + 0) = make_float4(denc[0], denc[1], denc[2], denc[3]);
    *reinterpret_cast<float4*>(ws_denc + i * 16 + 4) = make_float4(denc[4], denc[5], denc[6], denc[7]);
    *reinterpret_cast<float4*>(ws_denc + i * 16 + 8) = make_float4(denc[8], denc[9], 0.0f, 0.0f);
    ws_dout[i] = d_out;
  }
}

// [P][16] x4 GEMM operands + [P] d_out, then the scatter's replica scratch (256-B aligned)
static inline int64_t prop_scratch_offset(int64_t P) { return ((P * (16 + 16 + 16 + 16 + 1) * (int64_t)sizeof(float) + 1024 + 255) / 256) * 256; }

extern "C" int64_t tn_prop_workspace_bytes(int64_t num_points) {
  if (num_points < 0) return TN_EINVAL;
  return prop_scratch_offset(num_points) + tn_scatter_scratch_bytes(num_points, PL);
}

extern "C" int tn_prop_density_bwd(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins,
                                   const float* d_density, int64_t N, int32_t S, void* workspace, float* d_origins, float* d_directions,
                                   tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(net && origins && directions && e_bins && d_density && workspace, "tn_prop_density_bwd: null pointer");
  TN_REQUIRE(net->grid.table && net->grid.table_grad && net->w0 && net->b0 && net->w1 && net->b1 && net->gw0 && net->gb0 && net->gw1 && net->gb1,
             "tn_prop_density_bwd: null parameter/gradient pointer");
  TN_REQUIRE((d_origins == nullptr) == (d_directions == nullptr), "tn_prop_density_bwd: d_origins and d_directions must both be given or both NULL");
  TN_REQUIRE(net->grid.num_levels == PL, "tn_prop_density_bwd: proposal grids are built for %d levels, got %d", PL, net->grid.num_levels);
  TN_REQUIRE(net->grid.log2_hashmap_size >= 1 && net->grid.log2_hashmap_size <= 24, "tn_prop_density_bwd: bad log2_hashmap_size");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_prop_density_bwd: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(((uintptr_t)workspace % 256) == 0, "tn_prop_density_bwd: workspace must be 256-byte aligned");
  if (N == 0) return TN_OK;
  PropK k{make_gridk(net->grid), net->w0, net->b0, net->w1, net->b1, net->gw0, net->gb0, net->gw1, net->gb1};
  int64_t P = N * (int64_t)S;
  float* ws = reinterpret_cast<float*>(workspace);
  float *ws_da = ws, *ws_h = ws + P * 16, *ws_enc = ws + P * 32, *ws_denc = ws + P * 48, *ws_dout = ws + P * 64;
  int grid = (int)std::min<int64_t>(tn_cdiv(P, 256), 256 * 16);
  hipLaunchKernelGGL(k_prop_bwd_mlp, dim3(grid), dim3(256), 0, tn_s(stream), k, origins, directions, e_bins, d_density, N, S, ws_da, ws_dout, ws_enc,
                     ws_h, ws_denc);
  TN_CHECK_LAUNCH("tn_prop_density_bwd");
  // the two weight-gradient GEMMs run beside the (atomic-bound) table scatter on the companion stream
  hipStream_t st = tn_s(stream), side = tn_fork(st);
  int rcw = tn_wgrad_launch2(ws_da, 16, 16, ws_enc, 16, PF, net->gw0, PF, net->gb0, ws_dout, 1, 1, ws_h, 16, 16, net->gw1, 16, net->gb1, P, side ? side : st);
  int rcs = tn_grid_scatter_launch(net->grid, origins, directions, e_bins, ws_denc, 16, N, S, d_origins, d_directions,
                                   reinterpret_cast<char*>(workspace) + prop_scratch_offset(P), st);
  if (side) tn_join(st, side);
  return rcw ? rcw : rcs;
}
