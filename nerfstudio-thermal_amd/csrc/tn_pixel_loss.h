// Pixel terms of ThermalNerfactoModel.get_loss_dict as a device function, shared by tn_pixel_losses (tn_misc.hip) and the fused loss launch
// tn_train_losses (tn_sampler.hip).  `bid` / `nblk` stand in for blockIdx.x / gridDim.x so that the body can run as a slice of a larger grid.
#pragma once
#include "tn_common.h"
#include <algorithm>

// ------------------------------------------------------------------------------------------------ pixel losses
// losses_out[0..3] += {rgb_loss, thermal_loss, tv_pixel_loss, cross_channel_loss}; losses_out[4] (scratch) = number of RGB rays.
// Every block counts the RGB rays itself (N floats, a few KB): no separate count kernel / memset in front of the loss kernel.
__device__ __forceinline__ float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// one thread per ray, one quad of lanes per 2x2 patch (4 consecutive rays; is_thermal is constant inside a patch: PatchPixelSampler(patch_size=2))
__device__ __forceinline__ void pixel_losses_body(const float* __restrict__ pred_rgb, int rs, const float* __restrict__ pred_th, int ts,
                                                  const float* __restrict__ image, const float* __restrict__ is_thermal, int64_t N,
                                                  float thermal_mult, float tv_mult, float cross_mult, float* __restrict__ losses,
                                                  float* __restrict__ d_rgb, float* __restrict__ d_th, int bid, int nblk) {
  float l_rgb = 0.f, l_th = 0.f, l_tv = 0.f, l_cc = 0.f;
  __shared__ float sh_cnt[4];
  {
    // 16-byte loads, all requested before the first is consumed (a scalar loop here compiled to one load + wait per trip: N / 256 L2 round
    // trips in a row, 10 us in front of everything else this slice does)
    float cnt = 0.0f;  // exact: a count of at most 2^24 ones
    int64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(is_thermal) & 15) == 0) {
      const float4* it4 = reinterpret_cast<const float4*>(is_thermal);
      const int64_t n4 = N / 4;
      for (int64_t i0 = 0; i0 < n4; i0 += 4 * (int64_t)blockDim.x) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t i = i0 + (int64_t)u * blockDim.x + threadIdx.x;
          v[u] = make_float4(1.f, 1.f, 1.f, 1.f);  // (thermal: not counted)
          if (i < n4) v[u] = it4[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          cnt += (v[u].x == 0.0f ? 1.0f : 0.0f) + (v[u].y == 0.0f ? 1.0f : 0.0f) + (v[u].z == 0.0f ? 1.0f : 0.0f) + (v[u].w == 0.0f ? 1.0f : 0.0f);
      }
      done = n4 * 4;
    }
    for (int64_t i = done + threadIdx.x; i < N; i += blockDim.x) cnt += (is_thermal[i] == 0.0f) ? 1.0f : 0.0f;
    cnt = tn_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) sh_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
  }
  float n_rgb_rays = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
  if (bid == 0 && threadIdx.x == 0) { losses[4] = n_rgb_rays; losses[5] = (float)N - n_rgb_rays; }  // ray counts per spectrum (PSNR metrics)
  float n_patches = n_rgb_rays / 4.0f;
  // one thread per RAY; the 4 consecutive lanes of a quad are the 2x2 patch (N and blockDim are multiples of 4, so quads are whole).  One thread
  // per patch kept 32 loaded values + 16 results live (143 registers: spills inside the 80-register loss launch) and used a quarter of the lanes.
  // Every load of the ray first (the gradient buffers may alias the predictions' buffer -- columns of one [N,4] array --, so the compiler keeps
  // each read-modify-write in program order: 40 round trips in a row when they are written where they are used), then the arithmetic, then the
  // accumulators: read together, added to in the order the terms come (pixel term, then the patch term), written.
  const int lane = threadIdx.x & 63, quad0 = lane & ~3, k = lane & 3;
  for (int64_t i = bid * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)nblk * blockDim.x) {
    const float th = is_thermal[i], nt = 1.0f - th;
    const float p = pred_th[i * ts];
    float imgv[3], prv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { imgv[c] = image[i * 3 + c]; prv[c] = pred_rgb[i * rs + c]; }
    float add_r[3], gsum = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float img = imgv[c];
      float gt = img * nt;  // rgb_to_rgbt_image
      float a = gt * nt, b = prv[c] * nt;
      float df = a - b;
      l_rgb += df * df;
      add_r[c] = -2.0f * df * nt / ((float)N * 3.0f);
      gsum += gt;
    }
    const float grey_own = gsum / 3.0f;
    float add_t;
    {
      float gt_t = imgv[0] * th;
      float a = gt_t * th, b = p * th;
      float df = a - b;
      l_th += df * df;
      add_t = thermal_mult * (-2.0f * df * th) / (float)N;
    }
    // the patch: is_thermal of the four rays, their thermal predictions and grey values
    int any_th = th != 0.0f ? 1 : 0;
    any_th |= __shfl_xor(any_th, 1, 64);
    any_th |= __shfl_xor(any_th, 2, 64);
    float pt[4], grey[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { pt[j] = __shfl(p, quad0 + j, 64); grey[j] = __shfl(grey_own, quad0 + j, 64); }
    float dp_own = 0.0f;
    const bool patch_terms = !any_th && n_patches > 0.0f;
    if (patch_terms) {
      // tv: |p0-p1| + |p0-p2| + |p1-p3| + |p2-p3| ; cross: |(p1-p0)-(g1-g0)| + |(p2-p0)-(g2-g0)| + |(p3-p1)-(g3-g1)| + |(p3-p2)-(g3-g2)|
      const int A[4] = {1, 2, 3, 3}, B[4] = {0, 0, 1, 2};
      float dp[4] = {0.f, 0.f, 0.f, 0.f};
      float tvw = tv_mult * 0.25f / n_patches, ccw = cross_mult * 0.25f / n_patches;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dpt = pt[A[e]] - pt[B[e]];
        float dg = grey[A[e]] - grey[B[e]];
        if (k == 0) { l_tv += fabsf(dpt); l_cc += fabsf(dpt - dg); }  // (the patch's sums once)
        float s = tvw * sgn(dpt) + ccw * sgn(dpt - dg);  // d|B-A| = d|A-B|
        dp[A[e]] += s;
        dp[B[e]] -= s;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j == k) dp_own = dp[j];
    }
    float acr[3], act = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) acr[c] = d_rgb ? d_rgb[i * rs + c] : 0.0f;
    if (d_th) act = d_th[i * ts];
    if (d_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) d_rgb[i * rs + c] = acr[c] + add_r[c];
    }
    if (d_th) {
      float t = act + add_t;
      if (patch_terms) t += dp_own;
      d_th[i * ts] = t;
    }
  }
  l_rgb = tn_wave_sum(l_rgb); l_th = tn_wave_sum(l_th); l_tv = tn_wave_sum(l_tv); l_cc = tn_wave_sum(l_cc);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&losses[0], l_rgb / ((float)N * 3.0f));
    atomicAdd(&losses[1], thermal_mult * l_th / (float)N);
    if (n_patches > 0.0f) {
      atomicAdd(&losses[2], tv_mult * 0.25f * l_tv / n_patches);
      atomicAdd(&losses[3], cross_mult * 0.25f * l_cc / n_patches);
    }
  }
}

static inline int pixel_loss_blocks(int64_t N) { return (int)std::min<int64_t>(tn_cdiv(N, 256), 256); }
