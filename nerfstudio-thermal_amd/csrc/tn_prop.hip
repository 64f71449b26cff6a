// Proposal-network density: one lane per sample point, everything fused (position -> contraction -> selector ->
// 5-level hash gather + trilinear -> Linear(10,16) ReLU Linear(16,1) -> trunc_exp -> *selector).
// Replaces Field.density_fn -> HashMLPDensityField.get_density (fields/base_field.py:48-68, fields/density_fields.py:95-118).
// Consecutive lanes are consecutive samples of one ray, so neighbouring lanes hit the same / adjacent grid cells on the
// coarse levels (gathers coalesce in the texture-address unit) and the 772 B of MLP weights are wave-uniform scalar loads.
#include "tn_common.h"
#include "tn_prop_point.h"

// SAVE_ENC (training iterations in which the proposal networks take a gradient): the 10 encoding features of every sample are kept
// (level-major [5][P] float2) so that k_prop_bwd_mlp does not repeat the 40 gathers per sample.
template <bool SAVE_ENC>
__global__ void __launch_bounds__(256, 2) k_prop_fwd(PropK net, const float* __restrict__ origins, const float* __restrict__ directions,
                                                     const float* __restrict__ e_bins, int64_t N, int S, float* __restrict__ density,
                                                     float* __restrict__ enc_out) {
  __shared__ __attribute__((aligned(16))) float s_w[PH * PROP_WROW + 4];
  prop_stage_weights(net, s_w);
  int64_t P = N * (int64_t)S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t ray = tn_div_index(i, S, P);
    int s = (int)(i - ray * S);
    const float* o = origins + ray * 3;
    const float* d = directions + ray * 3;
    const float* eb = e_bins + ray * (S + 1) + s;
    Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], eb[0], eb[1]);
    // (prop_density_sample of tn_prop_point.h, written out: called as a function the <false> instantiation is scheduled into 66 registers instead of
    // 38 -- the two must stay the same arithmetic, tests/test_datamanager_gpu.py compares their results bit for bit)
    float enc[PF];
#pragma unroll
    for (int l = 0; l < PL; ++l) {
      float2 v = tn_encode_level(net.g.table, c.px, c.py, c.pz, net.g.res[l], net.g.mask, (uint32_t)l * net.g.tsize);
      enc[2 * l] = v.x;
      enc[2 * l + 1] = v.y;
      if (SAVE_ENC) *reinterpret_cast<float2*>(enc_out + ((int64_t)l * P + i) * 2) = v;  // level-major [PL][P] float2: 512 contiguous bytes per wave and level
    }
    const float out = prop_mlp(s_w, enc);
    density[i] = c.sel ? expf(out) : 0.0f * expf(out);  // exp(x) * selector (0*inf = nan kept as torch would)
  }
}

extern "C" int tn_prop_density_fwd(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins, int64_t N,
                                   int32_t S, float* density, tn_stream_t stream) {
  return tn_prop_density_fwd_ex(net, origins, directions, e_bins, N, S, density, nullptr, stream);
}
// enc_out: NULL, or N*S*10 floats (level-major [5][N*S] float2) that receive the samples' encodings (for tn_prop_density_bwd_ex; csrc/tn_pipeline.hip)
int tn_prop_density_fwd_ex(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins, int64_t N, int32_t S,
                           float* density, float* enc_out, tn_stream_t stream) {
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
  if (enc_out != nullptr)
    hipLaunchKernelGGL(k_prop_fwd<true>, dim3(grid), dim3(256), 0, tn_s(stream), k, origins, directions, e_bins, N, S, density, enc_out);
  else
    hipLaunchKernelGGL(k_prop_fwd<false>, dim3(grid), dim3(256), 0, tn_s(stream), k, origins, directions, e_bins, N, S, density, enc_out);
  TN_CHECK_LAUNCH("tn_prop_density_fwd");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ backward
// k_prop_bwd_mlp: recompute the forward, then  d_out = g * exp(clamp(out,-15,15)) * sel  (trunc_exp backward),
//   d a_j = d_out*w1_j*[a_j>0],  d enc_k = sum_j d a_j w0_jk  -> ws_denc (level-major) for the table scatter (tn_scatter.hip), and the WEIGHT
//   GRADIENTS dW0 = dA^T ENC, db0 = sum dA, dW1 = dOUT^T H, db1 = sum dOUT in the same kernel (layout and products: see the kernel).
//   History: round 1 wrote 65 floats per sample to HBM for a separate batched GEMM; round 2 fused the weight gradients (MFMA, K = samples) but
//   kept the MLP on the vector ALU with 193 wave-uniform weights in VGPRs (248 registers, 2 waves per SIMD) and closed with 1024 atomics per
//   weight: 105 us per launch on average, 57 of them that burst.  Now 56 us.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// k_prop_bwd_mlp, all three matrix products on v_mfma_f32_16x16x4_f32 (D[i][n] = sum_k A[i][k] B[k][n]; lane l supplies A[l % 16][l / 16] and
// B[l / 16][l % 16], register r of lane l receives D[4 (l / 16) + r][l % 16]).  With g = l / 16, n = l % 16, and the wave's 64 samples in four
// blocks c of 16:
//   stage   lane = sample writes its row E[s] = [enc(10) | 1 | 0] (row stride 12 floats) and dd[s] = d_density * selector
//   forward T_c[j][s] = sum_f W0aug[j][f] E[16 c + s][f]   A = the weights (3 registers per lane hold [w0 | b0] for good), B = E from LDS.
//           Lane (g, n) receives the pre-activations of the hidden units 4 g + r, r = 0..3, of sample 16 c + n.
//   out     = b1 + sum_j w1[j] relu(T[j][s]): 4 terms in the lane, then the four rows g are added (two cross-row exchanges)
//   d_out   = dd * exp(clamp(out)) (one exp per sample and row),  dA[j][s] = [T > 0] d_out w1[j]  -- in the registers of lane (g, n), which is
//           exactly operand B of
//   d enc   X_c[f][s] = sum_j W0[j][f] dA[j][16 c + s]   K = j = 4 g + r: straight from the registers, A = 4 registers of weights.  Lane (g, n)
//           receives features 4 g .. 4 g + 3 = levels 2 g, 2 g + 1 of sample 16 c + n: stored level-major (128-B runs).
//   weights G[f][j] += sum_s E[s][f] dA[j][s]   K = s: dA goes through LDS once ([s][j] rows of 16 floats) and comes back with the samples on
//           the k index, beside E^T; row 10 = db0
//   dW1[j], db1: per-lane sums over the loop, added up over the lanes at the end.
// The weights live in 11 registers per lane instead of 193 wave-uniform values (which the compiler kept in VGPRs: 248 of them, 2 waves per SIMD).
#define PB_RS 12   // floats per row of E ([enc(10) | 1 | 0]): 16 rows read at one column -> banks 12 n + g: every bank twice (the floor for 64 lanes)
#define PB_RS2 16  // floats per row of dA: ds_write_b128 at 16 s + 4 g and ds_read_b32 at 16 (4 t + g) + n are both conflict-free
#define PB_WAVE_FLOATS (64 * PB_RS + 64 * PB_RS2 + 64)
#define PB_WAVE_SYNC()                                    \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

__device__ __forceinline__ float prop_row_sum(float x) {  // sum over the 16 lanes of a DPP row, result in every lane
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));  // row_half_mirror
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));  // row_mirror
  return x;
}

// One block of 16 waves per CU (4 per SIMD): the kernel ends with one atomic per weight and block, and atomics on one 64-B line execute one after
// the other (~25 ns each).  With 1024 blocks of 4 waves that burst was 57 of the kernel's 105 us.
#define PB_THREADS 1024
// HAVE_ENC: the forward of this iteration kept the samples' encodings (k_prop_fwd<true>): they are read back (five coalesced float2 per lane) instead of
// gathered again (40 table reads per sample: ~2/5 of this kernel) -- the same values, bit for bit.
template <bool HAVE_ENC>
__global__ void __launch_bounds__(PB_THREADS) k_prop_bwd_mlp(PropK net, const float* __restrict__ origins, const float* __restrict__ directions,
                                                               const float* __restrict__ e_bins, const float* __restrict__ d_density, int64_t N, int S,
                                                               float* __restrict__ ws_denc, uint32_t* __restrict__ zero_ptr, int zero_words,
                                                               const float* __restrict__ saved_enc) {
  __shared__ __attribute__((aligned(16))) float lds[(PB_THREADS / 64) * PB_WAVE_FLOATS];
  tn_zero_words(zero_ptr, zero_words);  // the bucket counters of the scatter that follows on this stream
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane >> 4, n = lane & 15;
  float* E = lds + wv * PB_WAVE_FLOATS;
  float* DA = E + 64 * PB_RS;
  float* DD = DA + 64 * PB_RS2;
  float wf[3], wb[4], w1q[4];
#pragma unroll
  for (int t = 0; t < 3; ++t) {  // forward operand A[i = hidden n][k = feature 4 t + g]
    const int f = 4 * t + g;
    wf[t] = f < PF ? net.w0[n * PF + f] : (f == PF ? net.b0[n] : 0.0f);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    wb[r] = n < PF ? net.w0[(4 * g + r) * PF + n] : 0.0f;  // d enc operand A[i = feature n][k = g <-> hidden 4 g + r] of k-step r
    w1q[r] = net.w1[4 * g + r];
  }
  const float b1 = net.b1[0];
  int64_t P = N * (int64_t)S;
  // Two independent accumulators for the weight-gradient product: 16 MFMAs that accumulate into ONE register quad wait for each other (the
  // pipe's depth, 16 times over); two interleaved chains of eight halve that (four chains spill at the 128 registers of 4 waves per SIMD).
  f32x4_t G[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float dw1[4] = {0.f, 0.f, 0.f, 0.f}, sum_dout = 0.0f;
  // A trip = 64 samples.  The wave is ONE in-order instruction stream and each trip used to be a chain: global loads -> LDS -> MFMAs ->
  // shuffles -> exp -> MFMAs -> stores -> LDS -> MFMAs; with ~4 trips per wave at level 0 the loads' latency was paid four times over
  // (66 us for 1 M samples against ~13 us of vector / matrix work).  Now the NEXT trip's inputs are requested before this trip's matrix work
  // (11 registers), and the four 16-sample blocks of a trip run as four interleaved chains instead of one after the other.
  float n_enc[PF], n_dd = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto fetch = [&](int64_t b0) {
    const int64_t i = b0 + lane;
    const bool live = i < P;
    const int64_t ic = live ? i : P - 1;
    int64_t ray = tn_div_index(ic, S, P);
    int s = (int)(ic - ray * S);
    const float* o = origins + ray * 3;
    const float* d = directions + ray * 3;
    const float* eb = e_bins + ray * (S + 1) + s;
    Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], eb[0], eb[1]);
    n_dd = (live && c.sel) ? d_density[ic] : 0.0f;
#pragma unroll
    for (int l = 0; l < PL; ++l) {
      const float2 v = HAVE_ENC ? *reinterpret_cast<const float2*>(saved_enc + ((int64_t)l * P + ic) * 2)
                                : tn_encode_level(net.g.table, c.px, c.py, c.pz, net.g.res[l], net.g.mask, (uint32_t)l * net.g.tsize);
      n_enc[2 * l] = v.x;
      n_enc[2 * l + 1] = v.y;
    }
  };
  int64_t base = (blockIdx.x * (int64_t)(blockDim.x >> 6) + wv) * 64;
  if (base < P) fetch(base);
  for (; base < P; base += stride) {
    *reinterpret_cast<float4*>(E + lane * PB_RS + 0) = make_float4(n_enc[0], n_enc[1], n_enc[2], n_enc[3]);
    *reinterpret_cast<float4*>(E + lane * PB_RS + 4) = make_float4(n_enc[4], n_enc[5], n_enc[6], n_enc[7]);
    *reinterpret_cast<float4*>(E + lane * PB_RS + 8) = make_float4(n_enc[8], n_enc[9], 1.0f, 0.0f);
    DD[lane] = n_dd;
    if (base + stride < P) fetch(base + stride);  // in flight during everything below
    PB_WAVE_SYNC();
    // ---- the four 16-sample blocks c of the trip, two at a time as two interleaved chains (four at a time spill at 128 registers)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // forward: pre-activations of the hidden units 4 g + r of sample 16 c + n
      f32x4_t T[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          T[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], E[(16 * (2 * h + q) + n) * PB_RS + 4 * t + g], T[q], 0, 0, 0);
      float part[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        part[q] = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[q] = fmaf(w1q[r], fmaxf(T[q][r], 0.0f), part[q]);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) part[q] += __shfl_xor(part[q], 16, 64);  // the four rows hold the four quarters of the hidden layer
#pragma unroll
      for (int q = 0; q < 2; ++q) part[q] += __shfl_xor(part[q], 32, 64);
      f32x4_t X[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      float da[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = 2 * h + q;
        const float d_out = DD[16 * c + n] * expf(fminf(fmaxf(b1 + part[q], -15.0f), 15.0f));  // trunc_exp backward (the clamped exp is finite: 0 stays 0)
        if (g == 0) sum_dout += d_out;  // (every row computes it: one of them counts)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dw1[r] = fmaf(d_out, fmaxf(T[q][r], 0.0f), dw1[r]);
          da[q][r] = T[q][r] > 0.0f ? d_out * w1q[r] : 0.0f;
        }
      }
      // d enc of sample 16 c + n, features 4 g .. 4 g + 3: K = hidden, k-step r <-> hidden 4 g + r, straight from the registers
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) X[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[r], da[q][r], X[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = 2 * h + q;
        // dA[s][j] rows for the weight-gradient product
        *reinterpret_cast<float4*>(DA + (16 * c + n) * PB_RS2 + 4 * g) = make_float4(da[q][0], da[q][1], da[q][2], da[q][3]);
        const int64_t is = base + 16 * c + n;
        if (is < P && g < 3) {  // level-major [PL][P] float2: 16 lanes write 128 consecutive bytes (the bin pass reads them the same way)
          *reinterpret_cast<float2*>(ws_denc + ((int64_t)(2 * g) * P + is) * 2) = make_float2(X[q][0], X[q][1]);
          if (g < 2) *reinterpret_cast<float2*>(ws_denc + ((int64_t)(2 * g + 1) * P + is) * 2) = make_float2(X[q][2], X[q][3]);
        }
      }
    }
    PB_WAVE_SYNC();
    // ---- dW0 | db0 (transposed: [feature][hidden]): K = the wave's 64 samples, four at a time; chain q takes the k-steps t = q (mod 2)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float e = n < PB_RS ? E[(4 * t + g) * PB_RS + n] : 0.0f;
      G[t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(e, DA[(4 * t + g) * PB_RS2 + n], G[t & 1], 0, 0, 0);
    }
    PB_WAVE_SYNC();  // E, dA and dd have been read: the next trip may write them
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) G[0][r] = G[0][r] + G[1][r];
  // ---- block-level sum: every wave leaves its partial sums in its own LDS region, 273 threads add them up, one atomic per weight and block.
  // register r of lane (g, n) = G[feature 4 g + r][hidden n];  dw1[r] of lane (g, n) = this lane's samples' share of dW1[4 g + r]
#pragma unroll
  for (int r = 0; r < 4; ++r) dw1[r] = prop_row_sum(dw1[r]);
  sum_dout = tn_wave_sum(sum_dout);
  {
    float* mine = E;  // [0..255] [feature][hidden] tile (rows 0..9 dW0^T, row 10 db0), [256..271] dW1, [272] db1
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[(4 * g + r) * 16 + n] = G[0][r];
    if (n == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[256 + 4 * g + r] = dw1[r];
    }
    if (lane == 0) mine[272] = sum_dout;
  }
  __syncthreads();
  if (threadIdx.x < 273) {
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < PB_THREADS / 64; ++w) v += lds[w * PB_WAVE_FLOATS + threadIdx.x];
    const int f = threadIdx.x >> 4, h = threadIdx.x & 15;
    if (v != 0.0f) {
      if (f < PF) atomicAdd(net.gw0 + h * PF + f, v);
      else if (f == PF) atomicAdd(net.gb0 + h, v);
      else if (f == 16) atomicAdd(net.gw1 + h, v);
      else if (threadIdx.x == 272) atomicAdd(net.gb1, v);
    }
  }
}

// d enc (level-major [5][P] float2; the region keeps its [P][16] size), then the scatter's scratch (256-B aligned)
static inline int64_t prop_scratch_offset(int64_t P) { return ((P * 16 * (int64_t)sizeof(float) + 1024 + 255) / 256) * 256; }

extern "C" int64_t tn_prop_workspace_bytes(int64_t num_points) {
  if (num_points < 0) return TN_EINVAL;
  return prop_scratch_offset(num_points) + tn_scatter_scratch_bytes(num_points, PL);
}

extern "C" int tn_prop_density_bwd(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins,
                                   const float* d_density, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins,
                                   float* d_directions, tn_stream_t stream) {
  return tn_prop_density_bwd_ex(net, origins, directions, e_bins, d_density, N, S, workspace, workspace_bytes, d_origins, d_directions, nullptr, stream);
}
// saved_enc: NULL, or what tn_prop_density_fwd_ex(enc_out) kept for these very samples and parameters
int tn_prop_density_bwd_ex(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins, const float* d_density, int64_t N,
                           int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins, float* d_directions, const float* saved_enc,
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
  TN_REQUIRE(workspace_bytes >= tn_prop_workspace_bytes(N * (int64_t)S), "tn_prop_density_bwd: workspace of %lld bytes, tn_prop_workspace_bytes(%lld) = %lld",
             (long long)workspace_bytes, (long long)(N * (int64_t)S), (long long)tn_prop_workspace_bytes(N * (int64_t)S));
  if (N == 0) return TN_OK;
  PropK k{make_gridk(net->grid), net->w0, net->b0, net->w1, net->b1, net->gw0, net->gb0, net->gw1, net->gb1};
  int64_t P = N * (int64_t)S;
  float* ws_denc = reinterpret_cast<float*>(workspace);
  // one block per CU: every wave makes ~4 trips at level 0 (1 M samples) and the weight gradients are flushed 256 times
  int grid = (int)std::min<int64_t>(tn_cdiv(P, PB_THREADS), 256);
  const int threads = PB_THREADS;
  hipStream_t st = tn_s(stream);
  void* scratch = reinterpret_cast<char*>(workspace) + prop_scratch_offset(P);
  uint32_t* zp;
  int zw;
  tn_grid_scatter_counters(net->grid, P, scratch, &zp, &zw);
  if (saved_enc != nullptr)
    hipLaunchKernelGGL(k_prop_bwd_mlp<true>, dim3(grid), dim3(threads), 0, st, k, origins, directions, e_bins, d_density, N, S, ws_denc, zp, zw, saved_enc);
  else
    hipLaunchKernelGGL(k_prop_bwd_mlp<false>, dim3(grid), dim3(threads), 0, st, k, origins, directions, e_bins, d_density, N, S, ws_denc, zp, zw, saved_enc);
  TN_CHECK_LAUNCH("tn_prop_density_bwd");
  return tn_grid_scatter_launch(net->grid, origins, directions, e_bins, ws_denc, TN_LD_LEVEL_MAJOR, N, S, d_origins, d_directions, scratch, st, nullptr,
                                zw > 0);
}
