// Pixel terms of ThermalNerfactoModel.get_loss_dict as a device function, shared by tn_pixel_losses (tn_misc.hip) and the fused loss launch
// tn_train_losses (tn_sampler.hip).  `bid` / `nblk` stand in for blockIdx.x / gridDim.x so that the body can run as a slice of a larger grid.
#pragma once
#include "tn_common.h"
#include <algorithm>

// ------------------------------------------------------------------------------------------------ pixel losses
// losses_out[0..3] += {rgb_loss, thermal_loss, tv_pixel_loss, cross_channel_loss}; losses_out[4] (scratch) = number of RGB rays.
// Every block counts the RGB rays itself (N floats, a few KB): no separate count kernel / memset in front of the loss kernel.
__device__ __forceinline__ float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// one thread per 2x2 patch (4 consecutive rays; is_thermal is constant inside a patch: PatchPixelSampler(patch_size=2))
__device__ __forceinline__ void pixel_losses_body(const float* __restrict__ pred_rgb, int rs, const float* __restrict__ pred_th, int ts,
                                                  const float* __restrict__ image, const float* __restrict__ is_thermal, int64_t N,
                                                  float thermal_mult, float tv_mult, float cross_mult, float* __restrict__ losses,
                                                  float* __restrict__ d_rgb, float* __restrict__ d_th, int bid, int nblk) {
  float l_rgb = 0.f, l_th = 0.f, l_tv = 0.f, l_cc = 0.f;
  __shared__ float sh_cnt[4];
  {
    float cnt = 0.0f;  // exact: a count of at most 2^24 ones
    for (int64_t i = threadIdx.x; i < N; i += blockDim.x) cnt += (is_thermal[i] == 0.0f) ? 1.0f : 0.0f;  // N/256 independent loads per thread
    cnt = tn_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) sh_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
  }
  float n_rgb_rays = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
  if (bid == 0 && threadIdx.x == 0) { losses[4] = n_rgb_rays; losses[5] = (float)N - n_rgb_rays; }  // ray counts per spectrum (PSNR metrics)
  float n_patches = n_rgb_rays / 4.0f;
  int64_t Q = N / 4;
  for (int64_t q = bid * (int64_t)blockDim.x + threadIdx.x; q < Q; q += (int64_t)nblk * blockDim.x) {
    float pt[4], grey[4];
    bool rgb_patch = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t i = q * 4 + k;
      float th = is_thermal[i], nt = 1.0f - th;
      if (th != 0.0f) rgb_patch = false;
      float gsum = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float img = image[i * 3 + c];
        float gt = img * nt;  // rgb_to_rgbt_image
        float a = gt * nt, b = pred_rgb[i * rs + c] * nt;
        float df = a - b;
        l_rgb += df * df;
        if (d_rgb) d_rgb[i * rs + c] += -2.0f * df * nt / ((float)N * 3.0f);
        gsum += gt;
      }
      grey[k] = gsum / 3.0f;
      float p = pred_th[i * ts];
      pt[k] = p;
      float gt_t = image[i * 3] * th;
      float a = gt_t * th, b = p * th;
      float df = a - b;
      l_th += df * df;
      if (d_th) d_th[i * ts] += thermal_mult * (-2.0f * df * th) / (float)N;
    }
    if (rgb_patch && n_patches > 0.0f) {
      // tv: |p0-p1| + |p0-p2| + |p1-p3| + |p2-p3| ; cross: |(p1-p0)-(g1-g0)| + |(p2-p0)-(g2-g0)| + |(p3-p1)-(g3-g1)| + |(p3-p2)-(g3-g2)|
      const int A[4] = {1, 2, 3, 3}, B[4] = {0, 0, 1, 2};
      float dp[4] = {0.f, 0.f, 0.f, 0.f};
      float tvw = tv_mult * 0.25f / n_patches, ccw = cross_mult * 0.25f / n_patches;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dpt = pt[A[e]] - pt[B[e]];
        float dg = grey[A[e]] - grey[B[e]];
        l_tv += fabsf(dpt);
        l_cc += fabsf(dpt - dg);
        float s = tvw * sgn(dpt) + ccw * sgn(dpt - dg);  // d|B-A| = d|A-B|
        dp[A[e]] += s;
        dp[B[e]] -= s;
      }
      if (d_th) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d_th[(q * 4 + k) * ts] += dp[k];
      }
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

static inline int pixel_loss_blocks(int64_t N) { return (int)std::min<int64_t>(tn_cdiv(N / 4, 256), 256); }
