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
//   d a_j = d_out*w1_j*[a_j>0],  d enc_k = sum_j d a_j w0_jk  -> ws_denc for the table scatter (tn_scatter.hip), and the WEIGHT GRADIENTS
//   dW0 = dA^T ENC, db0 = sum dA, dW1 = dOUT^T H, db1 = sum dOUT in the same kernel:
//   a wave transposes its 64 samples' dA [64][16] and ENC [64][16: cols 0-9 enc, col 10 = 1 -> the bias column] through LDS and feeds them
//   to v_mfma_f32_16x16x4_f32 (K = samples): written sample-major, the operand of k-step t is the LINEAR read lds[64 t + lane] (lane l holds
//   row/col l % 16 of sample 4 t + l / 16), so the transpose costs 8 ds_write_b128 + 32 ds_read_b32 per 64 samples.  The accumulators
//   (2 x 4 registers) persist over the wave's grid-stride loop; one block-level sum in LDS and one atomic burst per block close the kernel.
//   Before: 65 floats per sample written to HBM (273 MB at level 0) and read back by a separate batched GEMM launch on a companion stream.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_prop_bwd_mlp(PropK net, const float* __restrict__ origins, const float* __restrict__ directions,
                                                      const float* __restrict__ e_bins, const float* __restrict__ d_density, int64_t N, int S,
                                                      float* __restrict__ ws_denc, uint32_t* __restrict__ zero_ptr, int zero_words) {
  __shared__ float lds[4][2][64 * 16];  // per wave: operand A, operand B (8 KB)
  tn_zero_words(zero_ptr, zero_words);  // the bucket counters of the scatter that follows on this stream
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* bufA = lds[wv][0];
  float* bufB = lds[wv][1];
  int64_t P = N * (int64_t)S;
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float sum_dout = 0.0f;
  // every wave of the grid makes the same number of trips (inactive lanes contribute zeros): the wave-level LDS hand-over below needs all lanes
  for (int64_t base = (blockIdx.x * (int64_t)(blockDim.x >> 6) + wv) * 64; base < P; base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = base + lane;
    const bool live = i < P;
    const int64_t ic = live ? i : P - 1;
    int64_t ray = ic / S;
    int s = (int)(ic - ray * S);
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
    float d_out = (live && c.sel) ? d_density[ic] * expf(fminf(fmaxf(out, -15.0f), 15.0f)) : 0.0f;
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
    if (live) {
      *reinterpret_cast<float4*>(ws_denc + i * 16 + 0) = make_float4(denc[0], denc[1], denc[2], denc[3]);
      *reinterpret_cast<float4*>(ws_denc + i * 16 + 4) = make_float4(denc[4], denc[5], denc[6], denc[7]);
      *reinterpret_cast<float4*>(ws_denc + i * 16 + 8) = make_float4(denc[8], denc[9], 0.0f, 0.0f);
    }
    sum_dout += d_out;
    // ---- dW0 | db0: A = dA [sample][16], B = [enc(10) | 1 | 0...] [sample][16]
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(bufA + lane * 16 + 4 * q) = make_float4(da[4 * q], da[4 * q + 1], da[4 * q + 2], da[4 * q + 3]);
    *reinterpret_cast<float4*>(bufB + lane * 16 + 0) = make_float4(enc[0], enc[1], enc[2], enc[3]);
    *reinterpret_cast<float4*>(bufB + lane * 16 + 4) = make_float4(enc[4], enc[5], enc[6], enc[7]);
    *reinterpret_cast<float4*>(bufB + lane * 16 + 8) = make_float4(enc[8], enc[9], 1.0f, 0.0f);
    *reinterpret_cast<float4*>(bufB + lane * 16 + 12) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 16; ++t) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bufA[64 * t + lane], bufB[64 * t + lane], acc0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- dW1: A = [d_out | 0 ...] (row 0 of a 16-row operand), B = H [sample][16]
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(bufB + lane * 16 + 4 * q) = make_float4(hh[4 * q], hh[4 * q + 1], hh[4 * q + 2], hh[4 * q + 3]);
    bufA[lane] = d_out;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float av = (lane & 15) == 0 ? bufA[4 * t + (lane >> 4)] : 0.0f;
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bufB[64 * t + lane], acc1, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- block-level sum (plain LDS adds, one wave per turn: ds_add_f32 is lane-serialised on gfx950), then one atomic burst per block.
  // accumulator register r of lane l = entry [4 (l / 16) + r][l % 16] of the 16x16 result
  __syncthreads();
  float* red = &lds[0][0][0];  // [0..255] dW0|db0 tile, [256..271] dW1 (row 0 of its tile), [272] db1
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum_dout += __shfl_xor(sum_dout, o, 64);
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int idx = (4 * (lane >> 4) + r) * 16 + (lane & 15);
        red[idx] = (w == 0 ? 0.0f : red[idx]) + acc0[r];
      }
      if (lane < 16) red[256 + lane] = (w == 0 ? 0.0f : red[256 + lane]) + acc1[0];  // row 0 = register 0 of lanes 0..15
      if (lane == 0) red[272] = (w == 0 ? 0.0f : red[272]) + sum_dout;
    }
    __syncthreads();
  }
  {
    int j = threadIdx.x >> 4, k = threadIdx.x & 15;
    float v = red[threadIdx.x];
    if (v != 0.0f) {
      if (k < PF) atomicAdd(net.gw0 + j * PF + k, v);
      else if (k == PF) atomicAdd(net.gb0 + j, v);
    }
    if (threadIdx.x < 16) {
      float u = red[256 + threadIdx.x];
      if (u != 0.0f) atomicAdd(net.gw1 + threadIdx.x, u);
    }
    if (threadIdx.x == 16) {
      float u = red[272];
      if (u != 0.0f) atomicAdd(net.gb1, u);
    }
  }
}

// d enc [P][16], then the scatter's scratch (256-B aligned)
static inline int64_t prop_scratch_offset(int64_t P) { return ((P * 16 * (int64_t)sizeof(float) + 1024 + 255) / 256) * 256; }

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
  float* ws_denc = reinterpret_cast<float*>(workspace);
  // 1024 blocks: every wave makes ~4 trips at level 0 (1 M samples), so the weight-gradient accumulators are flushed 1024 times, not 16 384
  // (same-line global atomics serialise at ~25 ns each)
  int grid = (int)std::min<int64_t>(tn_cdiv(P, 256), 1024);
  hipStream_t st = tn_s(stream);
  void* scratch = reinterpret_cast<char*>(workspace) + prop_scratch_offset(P);
  uint32_t* zp;
  int zw;
  tn_grid_scatter_counters(net->grid, P, scratch, &zp, &zw);
  hipLaunchKernelGGL(k_prop_bwd_mlp, dim3(grid), dim3(256), 0, st, k, origins, directions, e_bins, d_density, N, S, ws_denc, zp, zw);
  TN_CHECK_LAUNCH("tn_prop_density_bwd");
  return tn_grid_scatter_launch(net->grid, origins, directions, e_bins, ws_denc, 16, N, S, d_origins, d_directions, scratch, st, nullptr, zw > 0);
}
