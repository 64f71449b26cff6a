// The launch that ends a training iteration's backward (k_pose_bwd_finish_check: pose gradient, loss sums, camera regulariser, GradScaler's
// non-finite check of what the table scatters do not see) as device functions: tn_misc.hip launches them as a kernel of their own; on iterations
// without a proposal update tn_scatter.hip runs them in the FIRST blocks of the main grid's fold launch (k_seg_fold) -- everything they read is
// final when the bin launch before the fold has ended, and nothing of the fold depends on them.
#pragma once
#include "tn_common.h"

// backward:  R = I + f1 K + f2 K^2, K = skew(v), K^2 = v v^T - |v|^2 I, theta = sqrt(clamp(|v|^2, 1e-4))
//   G = g_d (outer) d_in ;  dL/dt = g_o
//   dL/dv_m = f1 * skewpart(G)_m + f2 * ((G + G^T) v - 2 v tr G)_m + (<G,K> f1' + <G,K^2> f2') * dtheta/dn * 2 v_m
__device__ __forceinline__ void pose_bwd_body(const float* __restrict__ pose, const uint8_t* __restrict__ frozen, const int64_t* __restrict__ cam_idx,
                                              const float* __restrict__ d_in, const float* __restrict__ g_o, const float* __restrict__ g_d, int64_t N,
                                              int C, float* __restrict__ grad_pose, int bid, int nblk, float* __restrict__ nonfinite = nullptr) {
  int lane = tn_lane();
  bool bad = false;  // a non-finite contribution makes the pose gradient non-finite (GradScaler's found_inf of the camera optimiser's group)
  int64_t stride = (int64_t)nblk * blockDim.x;
  int64_t iters = tn_cdiv(N, stride);
  for (int64_t it = 0; it < iters; ++it) {
    int64_t i = it * stride + bid * (int64_t)blockDim.x + threadIdx.x;
    bool live = i < N;
    int64_t ii = live ? i : N - 1;
    int64_t cam = cam_idx[ii];
    if (cam < 0 || cam >= C) cam = 0;
    float out[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live && !(frozen != nullptr && frozen[cam])) {
      const float* p = pose + cam * 6;
      float v[3] = {p[3], p[4], p[5]};
      float n = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      float th = sqrtf(fmaxf(n, 1e-4f));
      float sn = sinf(th), cs = cosf(th);
      float f1 = sn / th, f2 = (1.0f - cs) / (th * th);
      float df1 = (th * cs - sn) / (th * th);
      float df2 = (th * sn - 2.0f * (1.0f - cs)) / (th * th * th);
      float dth_dn = (n >= 1e-4f) ? 0.5f / th : 0.0f;
      float gd[3] = {g_d[ii * 3], g_d[ii * 3 + 1], g_d[ii * 3 + 2]};
      float dd[3] = {d_in[ii * 3], d_in[ii * 3 + 1], d_in[ii * 3 + 2]};
      float G[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) G[a * 3 + b] = gd[a] * dd[b];
      float K[9] = {0.f, -v[2], v[1], v[2], 0.f, -v[0], -v[1], v[0], 0.f};
      float GK = 0.f, GK2 = 0.f, tr = G[0] + G[4] + G[8];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          GK += G[a * 3 + b] * K[a * 3 + b];
          GK2 += G[a * 3 + b] * (v[a] * v[b] - (a == b ? n : 0.0f));
        }
      float sk[3] = {G[2 * 3 + 1] - G[1 * 3 + 2], G[0 * 3 + 2] - G[2 * 3 + 0], G[1 * 3 + 0] - G[0 * 3 + 1]};
      float common = (GK * df1 + GK2 * df2) * dth_dn * 2.0f;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        float Gv = G[m * 3] * v[0] + G[m * 3 + 1] * v[1] + G[m * 3 + 2] * v[2];
        float Gtv = G[m] * v[0] + G[3 + m] * v[1] + G[6 + m] * v[2];
        out[3 + m] = f1 * sk[m] + f2 * (Gv + Gtv - 2.0f * v[m] * tr) + common * v[m];
        out[m] = g_o[ii * 3 + m];
      }
    }
    int64_t c0 = __shfl(cam, 0, 64);
    if (__all(cam == c0)) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        float r = tn_wave_sum(out[q]);
        if (lane == 0 && r != 0.0f) atomicAdd(grad_pose + cam * 6 + q, r);
        bad = bad || ((r - r) != 0.0f);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (out[q] != 0.0f) atomicAdd(grad_pose + cam * 6 + q, out[q]);
        bad = bad || ((out[q] - out[q]) != 0.0f);
      }
    }
  }
  if (nonfinite != nullptr && __any(bad) && lane == 0) *nonfinite = 1.0f;
}

// Column sums of the loss lines of tn_train_losses into the loss vector + (pose != NULL) the camera regulariser: one block's work.
// ATOMIC_GRAD: other blocks of the same launch add into grad_pose at the same time (k_pose_bwd_finish).
template <bool ATOMIC_GRAD>
__device__ __forceinline__ void losses_finish_body(const float* __restrict__ lines, float* __restrict__ losses, const float* __restrict__ pose, int C,
                                                   float trans_pen, float rot_pen, float scale, float* __restrict__ reg_out,
                                                   float* __restrict__ grad_pose) {
  __shared__ float sh[16][17];
  const int t = threadIdx.x, k = t & 15, g = t >> 4;  // 16 groups of 16 slots; group g sums lines g, g + 16, ... (threads beyond 256: barriers only)
  if (lines != nullptr) {
    if (g < 16) {
      float acc = 0.0f;
      for (int l = g; l < TN_LOSS_LINES; l += 16) acc += lines[l * 16 + k];
      sh[g][k] = acc;
    }
    __syncthreads();
    if (t < 16) {
      float v = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += sh[q][t];
      if (v != 0.0f) losses[t] += v;
    }
    __syncthreads();  // reg_out may be one of the 16 slots
  }
  if (pose != nullptr) {
    float r = 0.0f;
    for (int c = t; c < C; c += blockDim.x) {
      const float* p = pose + c * 6;
      float nt = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      float nr = sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
      r += (nt * trans_pen + nr * rot_pen) * scale / (float)C;
      if (grad_pose != nullptr) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          // torch.norm backward: x / |x| (0 at the origin)
          if (nt > 0.0f) { float v = p[m] / nt * trans_pen * scale / (float)C; if (ATOMIC_GRAD) atomicAdd(&grad_pose[c * 6 + m], v); else grad_pose[c * 6 + m] += v; }
          if (nr > 0.0f) { float v = p[3 + m] / nr * rot_pen * scale / (float)C; if (ATOMIC_GRAD) atomicAdd(&grad_pose[c * 6 + 3 + m], v); else grad_pose[c * 6 + 3 + m] += v; }
        }
      }
    }
    r = tn_wave_sum(r);
    if ((t & 63) == 0) atomicAdd(reg_out, r);
  }
}
#define TN_ADAM_MAX_RANGES 8  // ranges per launch of the multi-range entry points (Adam, non-finite checks)
// tn_pose_bwd_finish + GradScaler's non-finite check of everything the table scatters do not see (tn_pose_bwd_finish_check): the pose gradient
// through its contributions (this launch produces it), and up to TN_ADAM_MAX_RANGES small gradient ranges -- MLP weights, embeddings: final
// by now -- in extra blocks, 4096 floats each.  Together with TnGrid::nonfinite_flag this replaces the pass of tn_grad_nonfinite_ranges over the
// whole gradient arena (78 MB, 12 us on the serial chain of a step).
struct SmallRanges { int64_t off[TN_ADAM_MAX_RANGES], cnt[TN_ADAM_MAX_RANGES]; int32_t flag[TN_ADAM_MAX_RANGES], first_block[TN_ADAM_MAX_RANGES + 1]; int32_t n; };
struct PoseFinishArgs {
  const float* pose; const uint8_t* frozen; const int64_t* cam_idx; const float* d_in; const float* g_o; const float* g_d; int64_t N; int C;
  float* grad_pose; const float* lines; float* losses; float trans_pen, rot_pen, scale; float* reg_out; int pose_blocks;
  const float* grads; SmallRanges sr; float* found_inf; int pose_flag;
  int total_blocks;  // pose_blocks + 1 + the small ranges' blocks
};
// block b of total_blocks (any block size that is a multiple of 64; at most the first 256 threads sum the loss lines)
__device__ __forceinline__ void pose_finish_body(const PoseFinishArgs& a, int b) {
  if (b < a.pose_blocks) { pose_bwd_body(a.pose, a.frozen, a.cam_idx, a.d_in, a.g_o, a.g_d, a.N, a.C, a.grad_pose, b, a.pose_blocks, a.found_inf + a.pose_flag); return; }
  if (b == a.pose_blocks) { losses_finish_body<true>(a.lines, a.losses, a.pose, a.C, a.trans_pen, a.rot_pen, a.scale, a.reg_out, a.grad_pose); return; }
  const int cb = b - a.pose_blocks - 1;
  int k = 0;
  for (int i = 1; i < a.sr.n; ++i)
    if (cb >= a.sr.first_block[i]) k = i;
  const int64_t lo = (int64_t)(cb - a.sr.first_block[k]) * 4096, hi = lo + 4096 < a.sr.cnt[k] ? lo + 4096 : a.sr.cnt[k];
  const float* base = a.grads + a.sr.off[k];
  bool bad = false;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { const float x = base[i]; bad = bad || ((x - x) != 0.0f); }
  if (__any(bad) && (threadIdx.x & 63) == 0) a.found_inf[a.sr.flag[k]] = 1.0f;
}
