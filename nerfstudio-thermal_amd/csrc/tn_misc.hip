// Ray generation, camera-pose correction, pixel-space losses, Adam, error plumbing.
#include "tn_common.h"
#include "tn_pose_finish.h"
#include <algorithm>
#include "tn_pixel_loss.h"
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void tn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tn_last_error(void) { return g_err; }
// ---- companion streams (tn_common.h: tn_fork / tn_join) ------------------------------------------------------------------
#include <map>
#include <mutex>
namespace {
struct Companion {
  hipStream_t side = nullptr;
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  bool forked = false;  // work has been forked to `side` since the last join: a join without it has nothing to wait for
};
std::mutex g_comp_mu;
// key: (device, caller stream, companion index).  Index 0: the field backward's d-position kernel; 1, 2: the proposal networks' backward in
// tn_render_rays_train_bwd.
struct CompKey {
  int dev; hipStream_t user; int idx;
  bool operator<(const CompKey& o) const { return dev != o.dev ? dev < o.dev : (user != o.user ? user < o.user : idx < o.idx); }
};
std::map<CompKey, Companion> g_comp;
Companion* companion_of(hipStream_t user, int idx) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_comp_mu);
  CompKey key{dev, user, idx};
  auto it = g_comp.find(key);
  if (it != g_comp.end()) return &it->second;
  if (getenv("TN_NO_FORK") != nullptr) return nullptr;  // debugging aid: everything on the caller's stream
  Companion c;
  // highest priority: the companion's kernels are short and finish while the scatter on the caller's stream keeps the LDS units busy;
  // at equal priority the scatter's thousands of resident blocks starve them and the join waits for a tail
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, idx == 0 ? hi : 0) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&c.fork_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c.join_ev, hipEventDisableTiming) != hipSuccess)
    return nullptr;
  return &(g_comp[key] = c);
}
}  // namespace
hipStream_t tn_fork_n(hipStream_t user, int idx) {
  Companion* c = companion_of(user, idx);
  if (c == nullptr) return nullptr;
  if (hipEventRecord(c->fork_ev, user) != hipSuccess || hipStreamWaitEvent(c->side, c->fork_ev, 0) != hipSuccess) return nullptr;
  c->forked = true;
  return c->side;
}
hipStream_t tn_fork(hipStream_t user) { return tn_fork_n(user, 0); }
void tn_join_n(hipStream_t user, int idx) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  Companion c;
  {
    std::lock_guard<std::mutex> lk(g_comp_mu);
    auto it = g_comp.find(CompKey{dev, user, idx});
    if (it == g_comp.end() || !it->second.forked) return;  // (an event record + wait on an idle companion is not free: two queue round trips)
    it->second.forked = false;
    c = it->second;
  }
  (void)hipEventRecord(c.join_ev, c.side);
  (void)hipStreamWaitEvent(user, c.join_ev, 0);
}
void tn_join_all(hipStream_t user) { tn_join_n(user, 0); }
void tn_join(hipStream_t user, hipStream_t companion) {
  Companion* c = companion_of(user, 0);
  if (c == nullptr || companion != c->side || !c->forked) return;
  c->forked = false;
  (void)hipEventRecord(c->join_ev, c->side);
  (void)hipStreamWaitEvent(user, c->join_ev, 0);
}

// Teardown of the only process-wide state the library keeps: the companion streams and their events (created lazily by tn_fork, one per
// (device, caller stream)).  Waits for the companions, destroys them, forgets them; a later call simply creates new ones.
extern "C" int tn_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_comp_mu);
  int cur = 0;
  (void)hipGetDevice(&cur);
  int rc = TN_OK;
  for (auto& kv : g_comp) {
    (void)hipSetDevice(kv.first.dev);
    Companion& c = kv.second;
    if (c.side != nullptr && hipStreamSynchronize(c.side) != hipSuccess) rc = TN_ELAUNCH;
    if (c.fork_ev) (void)hipEventDestroy(c.fork_ev);
    if (c.join_ev) (void)hipEventDestroy(c.join_ev);
    if (c.side) (void)hipStreamDestroy(c.side);
  }
  g_comp.clear();
  (void)hipSetDevice(cur);
  return rc;
}

extern "C" int tn_version(void) { return 307; }  // 307: tn_comm_* / tn_allreduce_grads; 306 (round 6): TnNextSampling, TnTrainStep::next_sampling / sampling_done; 305 (round 5): TnSampleRays, TnTrainStep::next_sample; 304: tn_render_losses_bwd; 303: tn_train_step (one call per training iteration), TN_FIELD_MAX_IMAGES; 302 (round 4): workspace_bytes behind every workspace pointer (a short buffer is TN_EINVAL, not an out-of-bounds write), tn_field_encode_plan; 301: TN_BWD_COUNTERS_CLEAN, proposal workspaces hand d enc over level-major; 300 (round 3): signatures of tn_hash_scatter_workspace_bytes / tn_field_density_fwd changed in round 2, AMP Adam + train-step entry points added

extern "C" int tn_fill_zero(void* ptr, int64_t bytes, tn_stream_t stream) {
  TN_REQUIRE(ptr != nullptr && bytes >= 0, "tn_fill_zero: bad argument");
  if (bytes == 0) return TN_OK;
  hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, tn_s(stream));
  if (e != hipSuccess) { tn_set_error("tn_fill_zero: %s", hipGetErrorString(e)); return TN_ELAUNCH; }
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ ray generation
// Cameras._generate_rays_from_coords for PERSPECTIVE cameras with OPENCV distortion
// (cameras/cameras.py:598-655,781-786,886-909; cameras/camera_utils.py:343-446,286-298).
__device__ __forceinline__ void undistort(float xd, float yd, const float (&k)[6], float& xo, float& yo) {
  const float k1 = k[0], k2 = k[1], k3 = k[2], k4 = k[3], p1 = k[4], p2 = k[5];
  float x = xd, y = yd;
  for (int it = 0; it < 10; ++it) {
    float r = x * x + y * y;
    float d = 1.0f + r * (k1 + r * (k2 + r * (k3 + r * k4)));
    float fx = d * x + 2.0f * p1 * x * y + p2 * (r + 2.0f * x * x) - xd;
    float fy = d * y + 2.0f * p2 * x * y + p1 * (r + 2.0f * y * y) - yd;
    float d_r = k1 + r * (2.0f * k2 + r * (3.0f * k3 + r * 4.0f * k4));
    float d_x = 2.0f * x * d_r;
    float d_y = 2.0f * y * d_r;
    float fx_x = d + d_x * x + 2.0f * p1 * y + 6.0f * p2 * x;
    float fx_y = d_y * x + 2.0f * p1 * x + 2.0f * p2 * y;
    float fy_x = d_x * y + 2.0f * p2 * y + 2.0f * p1 * x;
    float fy_y = d + d_y * y + 2.0f * p2 * x + 6.0f * p1 * y;
    float den = fy_x * fx_y - fx_x * fy_y;
    float xn = fx * fy_y - fy * fx_y;
    float yn = fy * fx_x - fx * fy_x;
    bool ok = fabsf(den) > 1e-3f;
    x = x + (ok ? xn / den : 0.0f);
    y = y + (ok ? yn / den : 0.0f);
  }
  xo = x; yo = y;
}

// ---- N2: patch pixel sampler + ground-truth gather (one thread per ray) ---------------------------------------------------------
struct SamplePixelsArgs {
  const float* images; const int64_t* image_offsets; const int32_t* heights; const int32_t* widths; const float* is_thermal;
  const int64_t* image_idx; int num_images; const float* u; int64_t N; int ps; int64_t rays_per_image;
  int64_t* ray_indices; float* image; float* is_thermal_out; int64_t* camera_indices;
};
// one ray in two halves, so that a caller can put every load of the ray in front of its first store (the outputs may alias nothing, but the
// compiler cannot know: written as one body the kernel was a chain of ten dependent round trips -- index, pixel channel, store, next channel, ...):
// sample_pixel_pick = which image / patch / pixel + the address of its ground truth (loads only); sample_pixel_store = the outputs.
struct PixelPick { int64_t cam, y, x; const float* px; float th; };
__device__ __forceinline__ PixelPick sample_pixel_pick(const SamplePixelsArgs& a, int64_t r) {
  const int pp = a.ps * a.ps;
  int64_t i = a.rays_per_image > 0 ? r / a.rays_per_image : a.num_images - 1;  // every image holds rays_per_image rays, the last one the rest
  if (i > a.num_images - 1) i = a.num_images - 1;
  int64_t local = r - i * a.rays_per_image;
  int64_t patch = local / pp;
  int k = (int)(local - patch * pp);
  int dy = k / a.ps, dx = k - dy * a.ps;
  const float* up = a.u + (i * (a.rays_per_image / pp) + patch) * 3;
  const float u1 = up[1], u2 = up[2];
  const int H = a.heights[i], W = a.widths[i];
  PixelPick p;
  p.cam = a.image_idx[i];
  p.th = a.is_thermal[i];
  const int64_t off = a.image_offsets[i];
  // torch: floor(rand * [1, H - ps, W - ps]) in fp32 (float32 x int64 promotes to float32), then + patch offsets
  p.y = (int64_t)floorf(u1 * (float)(H - a.ps) + (float)dy);
  p.x = (int64_t)floorf(u2 * (float)(W - a.ps) + (float)dx);
  p.px = a.images + off + (p.y * W + p.x) * 3;
  return p;
}
__device__ __forceinline__ void sample_pixel_store(const SamplePixelsArgs& a, int64_t r, const PixelPick& p, float c0, float c1, float c2) {
  a.ray_indices[r * 3 + 0] = p.cam;
  if (a.camera_indices != nullptr) a.camera_indices[r] = p.cam;
  a.ray_indices[r * 3 + 1] = p.y;
  a.ray_indices[r * 3 + 2] = p.x;
  a.image[r * 3 + 0] = c0;
  a.image[r * 3 + 1] = c1;
  a.image[r * 3 + 2] = c2;
  a.is_thermal_out[r] = p.th;
}
__global__ void k_sample_pixels(SamplePixelsArgs a) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < a.N; r += (int64_t)gridDim.x * blockDim.x) {
    const PixelPick p = sample_pixel_pick(a, r);
    const float c0 = p.px[0], c1 = p.px[1], c2 = p.px[2];
    sample_pixel_store(a, r, p, c0, c1, c2);
  }
}

static int sample_pixels_args(const char* who, const float* images, const int64_t* image_offsets, const int32_t* heights, const int32_t* widths,
                              const float* is_thermal, const int64_t* image_idx, int32_t num_images, const float* u, int64_t num_rays,
                              int32_t patch_size, int64_t* ray_indices, float* image, float* is_thermal_out, int64_t* camera_indices,
                              SamplePixelsArgs& a) {
  TN_REQUIRE(images && image_offsets && heights && widths && is_thermal && image_idx && u && ray_indices && image && is_thermal_out,
             "%s: null pointer", who);
  TN_REQUIRE(num_rays > 0 && num_images >= 1 && patch_size >= 1 && patch_size <= 8, "%s: bad num_rays=%lld num_images=%d patch_size=%d", who,
             (long long)num_rays, num_images, patch_size);
  const int64_t pp = (int64_t)patch_size * patch_size;
  const int64_t per = ((num_rays / num_images) / pp) * pp;  // rays of every image but the last
  const int64_t last = num_rays - (int64_t)(num_images - 1) * per;
  TN_REQUIRE(last > 0 && last % pp == 0, "%s: %lld rays over %d images do not split into whole %dx%d patches", who, (long long)num_rays,
             num_images, patch_size, patch_size);
  a = SamplePixelsArgs{images, image_offsets, heights, widths, is_thermal, image_idx, num_images, u, num_rays, patch_size, per,
                       ray_indices, image, is_thermal_out, camera_indices};
  return TN_OK;
}

extern "C" int tn_sample_pixels(const float* images, const int64_t* image_offsets, const int32_t* heights, const int32_t* widths,
                                const float* is_thermal, const int64_t* image_idx, int32_t num_images, const float* u, int64_t num_rays,
                                int32_t patch_size, int64_t* ray_indices, float* image, float* is_thermal_out, int64_t* camera_indices,
                                tn_stream_t stream) {
  if (num_rays == 0) return TN_OK;  // empty batches are valid and touch nothing
  SamplePixelsArgs a;
  int rc = sample_pixels_args("tn_sample_pixels", images, image_offsets, heights, widths, is_thermal, image_idx, num_images, u, num_rays, patch_size,
                              ray_indices, image, is_thermal_out, camera_indices, a);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sample_pixels, dim3((unsigned)std::min<int64_t>(tn_cdiv(num_rays, 256), 2048)), dim3(256), 0, tn_s(stream), a);
  TN_CHECK_LAUNCH("tn_sample_pixels");
  return TN_OK;
}

struct RaygenArgs {
  const float* c2w; const float* fx; const float* fy; const float* cx; const float* cy; const float* distortion;
  int any_distortion, num_cameras;
  float* origins; float* directions; float* pixel_area; float* directions_norm;
};
// one ray from (camera, pixel row, pixel column), in two halves like the pixel sampler: the camera's parameters (loads only), then the arithmetic
// and the stores
struct CamParams { float fx, fy, cx, cy, R[12], k[6]; };
__device__ __forceinline__ void raygen_load(const RaygenArgs& g, int64_t& cam, CamParams& c) {
  if (cam < 0 || cam >= g.num_cameras) cam = 0;  // host validates; keep the access in bounds regardless
  c.fx = g.fx[cam]; c.fy = g.fy[cam]; c.cx = g.cx[cam]; c.cy = g.cy[cam];
#pragma unroll
  for (int q = 0; q < 12; ++q) c.R[q] = g.c2w[cam * 12 + q];
#pragma unroll
  for (int q = 0; q < 6; ++q) c.k[q] = 0.0f;
  if (g.any_distortion) {  // (kernel-uniform; three 8-byte loads requested together)
    const float2* dk = reinterpret_cast<const float2*>(g.distortion + cam * 6);
    const float2 k01 = dk[0], k23 = dk[1], k45 = dk[2];
    c.k[0] = k01.x; c.k[1] = k01.y; c.k[2] = k23.x; c.k[3] = k23.y; c.k[4] = k45.x; c.k[5] = k45.y;
  }
}
__device__ __forceinline__ void raygen_compute(const RaygenArgs& g, int64_t i, const CamParams& cp, int64_t yi, int64_t xi) {
  float y = (float)yi + 0.5f;  // get_image_coords(pixel_offset=0.5)
  float x = (float)xi + 0.5f;
  const float fxc = cp.fx, fyc = cp.fy, cxc = cp.cx, cyc = cp.cy;
  // coord, coord_x_offset, coord_y_offset
  float u[3] = {(x - cxc) / fxc, (x - cxc + 1.0f) / fxc, (x - cxc) / fxc};
  float v[3] = {(y - cyc) / fyc, (y - cyc) / fyc, (y - cyc + 1.0f) / fyc};
  const float* R = cp.R;
  float dir[3][3];
  float nrm0 = 0.0f;
  const int nq = g.pixel_area ? 3 : 1;  // (the +1-pixel neighbours only serve pixel_area: two of the three Newton undistortions, the kernel's serial chain)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (q >= nq) break;
    float a = u[q], b = v[q];
    if (g.any_distortion) undistort(a, b, cp.k, a, b);
    b = -b;  // OpenCV -> OpenGL
    float dz = -1.0f;
    // sum(d[None,:] * R, dim=-1): row r of R dotted with d
    float w0 = a * R[0] + b * R[1] + dz * R[2];
    float w1 = a * R[4] + b * R[5] + dz * R[6];
    float w2 = a * R[8] + b * R[9] + dz * R[10];
    float n = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    n = fmaxf(n, 8.881784197001252e-16f);  // np.finfo(float).eps * 4
    dir[q][0] = w0 / n; dir[q][1] = w1 / n; dir[q][2] = w2 / n;
    if (q == 0) nrm0 = n;
  }
  g.origins[i * 3 + 0] = R[3]; g.origins[i * 3 + 1] = R[7]; g.origins[i * 3 + 2] = R[11];
  g.directions[i * 3 + 0] = dir[0][0]; g.directions[i * 3 + 1] = dir[0][1]; g.directions[i * 3 + 2] = dir[0][2];
  if (g.pixel_area) {
    float ddx = 0.0f, ddy = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float a = dir[0][c] - dir[1][c], b = dir[0][c] - dir[2][c];
      ddx += a * a; ddy += b * b;
    }
    g.pixel_area[i] = sqrtf(ddx) * sqrtf(ddy);
  }
  if (g.directions_norm) g.directions_norm[i] = nrm0;
}
__global__ void k_raygen(const int64_t* __restrict__ ray_indices, RaygenArgs g, int64_t N) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t cam = ray_indices[i * 3];
    const int64_t y = ray_indices[i * 3 + 1], x = ray_indices[i * 3 + 2];
    CamParams cp;
    raygen_load(g, cam, cp);
    raygen_compute(g, i, cp, y, x);
  }
}
// datamanager.next_train in one launch: pixel sampling + ground-truth gather + ray generation (the pixel goes from one to the other in registers).
// FOUR lanes per ray: lane q of the quad runs the Newton undistortion of coordinate q (the pixel, its +1-column and its +1-row neighbour, which
// only serve pixel_area; lane 3 idles) -- one thread per ray ran the three chains one after the other, ~2 000 dependent instructions on 64 waves
// that each have a SIMD to themselves.  Same expressions per coordinate as raygen_compute.
// ray r by the quad of lanes this lane belongs to (q = lane & 3); `store`: this lane writes the ray's outputs (a q == 0 lane).  Returns the ray
// (origin, direction, camera; the direction is valid in q == 0 lanes) for a caller that goes on with it (tn_next_sampling.h).
struct SampledRay { float o[3], d[3]; int64_t cam; };
__device__ __forceinline__ SampledRay sample_ray_quad(const SamplePixelsArgs& a, const RaygenArgs& g, int64_t r, int lane, bool store) {
  const int q = lane & 3, quad0 = lane & ~3;
  // two rounds of loads (index data; then the pixel and the camera's parameters together), then arithmetic and stores
  const PixelPick p = sample_pixel_pick(a, r);
  const float c0 = p.px[0], c1 = p.px[1], c2 = p.px[2];
  int64_t cam = p.cam;
  CamParams cp;
  raygen_load(g, cam, cp);
  if (store) sample_pixel_store(a, r, p, c0, c1, c2);
  const float y = (float)p.y + 0.5f, x = (float)p.x + 0.5f;  // get_image_coords(pixel_offset=0.5)
  // coord, coord_x_offset, coord_y_offset
  float ua = (q == 1) ? (x - cp.cx + 1.0f) / cp.fx : (x - cp.cx) / cp.fx;
  float vb = (q == 2) ? (y - cp.cy + 1.0f) / cp.fy : (y - cp.cy) / cp.fy;
  float dir[3] = {0.f, 0.f, 0.f}, nrm = 0.0f;
  const int nq = g.pixel_area ? 3 : 1;
  if (q < nq) {
    float aa = ua, bb = vb;
    if (g.any_distortion) undistort(aa, bb, cp.k, aa, bb);
    bb = -bb;  // OpenCV -> OpenGL
    const float dz = -1.0f;
    const float* R = cp.R;
    float w0 = aa * R[0] + bb * R[1] + dz * R[2];
    float w1 = aa * R[4] + bb * R[5] + dz * R[6];
    float w2 = aa * R[8] + bb * R[9] + dz * R[10];
    float n = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    n = fmaxf(n, 8.881784197001252e-16f);  // np.finfo(float).eps * 4
    dir[0] = w0 / n; dir[1] = w1 / n; dir[2] = w2 / n;
    nrm = n;
  }
  float d1[3], d2[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { d1[c] = __shfl(dir[c], quad0 + 1, 64); d2[c] = __shfl(dir[c], quad0 + 2, 64); }
  if (store) {
    g.origins[r * 3 + 0] = cp.R[3]; g.origins[r * 3 + 1] = cp.R[7]; g.origins[r * 3 + 2] = cp.R[11];
    g.directions[r * 3 + 0] = dir[0]; g.directions[r * 3 + 1] = dir[1]; g.directions[r * 3 + 2] = dir[2];
    if (g.pixel_area) {
      float ddx = 0.0f, ddy = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float da = dir[c] - d1[c], db = dir[c] - d2[c];
        ddx += da * da; ddy += db * db;
      }
      g.pixel_area[r] = sqrtf(ddx) * sqrtf(ddy);
    }
    if (g.directions_norm) g.directions_norm[r] = nrm;
  }
  SampledRay out;
  out.o[0] = cp.R[3]; out.o[1] = cp.R[7]; out.o[2] = cp.R[11];
  out.d[0] = dir[0]; out.d[1] = dir[1]; out.d[2] = dir[2];
  out.cam = p.cam;  // (as stored: the caller clamps it against ITS table, like the kernels that read camera_indices)
  return out;
}
__device__ __forceinline__ void sample_rays_body(const SamplePixelsArgs& a, const RaygenArgs& g, unsigned bid, unsigned nblk) {
  const int lane = threadIdx.x & 63;
  for (int64_t t = bid * (int64_t)blockDim.x + threadIdx.x; t < a.N * 4; t += (int64_t)nblk * blockDim.x)  // (whole quads: blockDim % 4 == 0)
    sample_ray_quad(a, g, t >> 2, lane, (lane & 3) == 0);
}
__global__ void k_sample_rays(SamplePixelsArgs a, RaygenArgs g) { sample_rays_body(a, g, blockIdx.x, gridDim.x); }

extern "C" int tn_raygen(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy, const float* cx, const float* cy,
                         const float* distortion, int32_t num_cameras, int64_t N, float* origins, float* directions, float* pixel_area,
                         float* directions_norm, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(ray_indices && c2w && fx && fy && cx && cy && origins && directions && pixel_area, "tn_raygen: null pointer");
  TN_REQUIRE(N >= 0 && num_cameras >= 1, "tn_raygen: bad N=%lld num_cameras=%d", (long long)N, num_cameras);
  RaygenArgs g{c2w, fx, fy, cx, cy, distortion, distortion != nullptr ? 1 : 0, num_cameras, origins, directions, pixel_area, directions_norm};
  hipLaunchKernelGGL(k_raygen, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, 256), 2048)), dim3(256), 0, tn_s(stream), ray_indices, g, N);
  TN_CHECK_LAUNCH("tn_raygen");
  return TN_OK;
}

extern "C" int tn_sample_rays(const float* images, const int64_t* image_offsets, const int32_t* heights, const int32_t* widths,
                              const float* is_thermal, const int64_t* image_idx, int32_t num_images, const float* u, int64_t num_rays,
                              int32_t patch_size, int64_t* ray_indices, float* image, float* is_thermal_out, int64_t* camera_indices,
                              const float* c2w, const float* fx, const float* fy, const float* cx, const float* cy, const float* distortion,
                              int32_t num_cameras, float* origins, float* directions, float* pixel_area, float* directions_norm,
                              tn_stream_t stream) {
  if (num_rays == 0) return TN_OK;  // empty batches are valid and touch nothing
  SamplePixelsArgs a;
  int rc = sample_pixels_args("tn_sample_rays", images, image_offsets, heights, widths, is_thermal, image_idx, num_images, u, num_rays, patch_size,
                              ray_indices, image, is_thermal_out, camera_indices, a);
  if (rc) return rc;
  TN_REQUIRE(c2w && fx && fy && cx && cy && origins && directions && num_cameras >= 1, "tn_sample_rays: bad camera arguments");  // (pixel_area may be NULL)
  RaygenArgs g{c2w, fx, fy, cx, cy, distortion, distortion != nullptr ? 1 : 0, num_cameras, origins, directions, pixel_area, directions_norm};
  hipLaunchKernelGGL(k_sample_rays, dim3((unsigned)std::min<int64_t>(tn_cdiv(num_rays * 4, 256), 4096)), dim3(256), 0, tn_s(stream), a, g);  // 4 lanes per ray
  TN_CHECK_LAUNCH("tn_sample_rays");
  return TN_OK;
}
// validated kernel arguments of a TnSampleRays block (shared by tn_sample_rays_args and the optimiser launch's co-work)
static int sample_rays_build(const char* who, const TnSampleRays* s, SamplePixelsArgs& a, RaygenArgs& g) {
  TN_REQUIRE(s != nullptr && s->num_rays > 0, "%s: null / empty TnSampleRays", who);
  int rc = sample_pixels_args(who, s->images, s->image_offsets, s->heights, s->widths, s->is_thermal, s->image_idx, s->num_images, s->u, s->num_rays,
                              s->patch_size, s->ray_indices, s->image, s->is_thermal_out, s->camera_indices, a);
  if (rc) return rc;
  TN_REQUIRE(s->c2w && s->fx && s->fy && s->cx && s->cy && s->origins && s->directions && s->num_cameras >= 1, "%s: bad camera arguments", who);
  g = RaygenArgs{s->c2w, s->fx, s->fy, s->cx, s->cy, s->distortion, s->distortion != nullptr ? 1 : 0, s->num_cameras, s->origins, s->directions, s->pixel_area,
                 s->directions_norm};
  return TN_OK;
}
extern "C" int tn_sample_rays_args(const TnSampleRays* s, tn_stream_t stream) {
  TN_REQUIRE(s != nullptr, "tn_sample_rays_args: null pointer");
  if (s->num_rays == 0) return TN_OK;
  SamplePixelsArgs a;
  RaygenArgs g;
  int rc = sample_rays_build("tn_sample_rays_args", s, a, g);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sample_rays, dim3((unsigned)std::min<int64_t>(tn_cdiv(s->num_rays * 4, 256), 4096)), dim3(256), 0, tn_s(stream), a, g);
  TN_CHECK_LAUNCH("tn_sample_rays_args");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ pose correction
struct Pose {
  float R[9];
  float t[3];
};
// exp_map_SO3xR3 (cameras/lie_groups.py:24-58)
__device__ __forceinline__ Pose pose_exp(const float* __restrict__ p) {
  Pose o;
  float v0 = p[3], v1 = p[4], v2 = p[5];
  float n = v0 * v0 + v1 * v1 + v2 * v2;
  float ang = sqrtf(fmaxf(n, 1e-4f));
  float inv = 1.0f / ang;
  float f1 = inv * sinf(ang);
  float f2 = inv * inv * (1.0f - cosf(ang));
  float K[9] = {0.f, -v2, v1, v2, 0.f, -v0, -v1, v0, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float kk = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
      o.R[i * 3 + j] = f1 * K[i * 3 + j] + f2 * kk + (i == j ? 1.0f : 0.0f);
    }
  o.t[0] = p[0]; o.t[1] = p[1]; o.t[2] = p[2];
  return o;
}

// one ray: prow = the camera's pose row, is_frozen = a non-trainable camera (identity)
__device__ __forceinline__ void pose_apply_ray(const float (&prow)[6], bool is_frozen, float ox, float oy, float oz, float dx, float dy, float dz,
                                               float (&o)[3], float (&d)[3]) {
  if (is_frozen) {
    // identity transform: origins + 0, bmm(I, d)
    o[0] = ox + 0.0f; o[1] = oy + 0.0f; o[2] = oz + 0.0f;
    d[0] = dx; d[1] = dy; d[2] = dz;
    return;
  }
  Pose p = pose_exp(prow);
  o[0] = ox + p.t[0]; o[1] = oy + p.t[1]; o[2] = oz + p.t[2];
  d[0] = p.R[0] * dx + p.R[1] * dy + p.R[2] * dz;
  d[1] = p.R[3] * dx + p.R[4] * dy + p.R[5] * dz;
  d[2] = p.R[6] * dx + p.R[7] * dy + p.R[8] * dz;
}
__device__ __forceinline__ void pose_fwd_body(const float* __restrict__ pose, const uint8_t* __restrict__ frozen, const int64_t* __restrict__ cam_idx,
                                              const float* __restrict__ o_in, const float* __restrict__ d_in, int64_t N, int C,
                                              float* __restrict__ o_out, float* __restrict__ d_out, int bid, int nblk) {
  for (int64_t i = bid * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)nblk * blockDim.x) {
    int64_t cam = cam_idx[i];
    if (cam < 0 || cam >= C) cam = 0;
    float ox = o_in[i * 3], oy = o_in[i * 3 + 1], oz = o_in[i * 3 + 2];
    float dx = d_in[i * 3], dy = d_in[i * 3 + 1], dz = d_in[i * 3 + 2];
    // (the camera's pose row is requested together with its frozen flag: one round trip, not two)
    float prow[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) prow[q] = pose[cam * 6 + q];
    float o[3], d[3];
    pose_apply_ray(prow, frozen != nullptr && frozen[cam], ox, oy, oz, dx, dy, dz, o, d);
    o_out[i * 3] = o[0]; o_out[i * 3 + 1] = o[1]; o_out[i * 3 + 2] = o[2];
    d_out[i * 3] = d[0]; d_out[i * 3 + 1] = d[1]; d_out[i * 3 + 2] = d[2];
  }
}
__global__ void k_pose_fwd(const float* __restrict__ pose, const uint8_t* __restrict__ frozen, const int64_t* __restrict__ cam_idx,
                           const float* __restrict__ o_in, const float* __restrict__ d_in, int64_t N, int C, float* __restrict__ o_out,
                           float* __restrict__ d_out) {
  pose_fwd_body(pose, frozen, cam_idx, o_in, d_in, N, C, o_out, d_out, blockIdx.x, gridDim.x);
}
// the two independent first steps of a training render in one launch: blockIdx.y = 0 the level-0 bins, 1 the pose correction of the rays
__global__ void k_pose_spaced_bins(const float* __restrict__ pose, const uint8_t* __restrict__ frozen, const int64_t* __restrict__ cam_idx,
                                   const float* __restrict__ o_in, const float* __restrict__ d_in, int64_t N, int C, float* __restrict__ o_out,
                                   float* __restrict__ d_out, int pose_blocks, const float* __restrict__ lin_bins,
                                   const float* __restrict__ jitter, const float* __restrict__ nears, const float* __restrict__ fars, int S,
                                   float* __restrict__ s_bins, float* __restrict__ e_bins) {
  if (blockIdx.y == 0) {
    tn_spaced_bins_body(lin_bins, jitter, nears, fars, N, S, s_bins, e_bins, blockIdx.x, gridDim.x);
  } else {
    if ((int)blockIdx.x >= pose_blocks) return;
    pose_fwd_body(pose, frozen, cam_idx, o_in, d_in, N, C, o_out, d_out, blockIdx.x, pose_blocks);
  }
}

extern "C" int tn_pose_apply_fwd(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* origins_in,
                                 const float* directions_in, int64_t N, int32_t num_cameras, float* origins_out, float* directions_out,
                                 tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(pose_adjustment && camera_indices && origins_in && directions_in && origins_out && directions_out, "tn_pose_apply_fwd: null pointer");
  TN_REQUIRE(N >= 0 && num_cameras >= 1, "tn_pose_apply_fwd: bad N=%lld C=%d", (long long)N, num_cameras);
  if (N == 0) return TN_OK;
  hipLaunchKernelGGL(k_pose_fwd, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, 256), 2048)), dim3(256), 0, tn_s(stream), pose_adjustment, frozen,
                     camera_indices, origins_in, directions_in, N, num_cameras, origins_out, directions_out);
  TN_CHECK_LAUNCH("tn_pose_apply_fwd");
  return TN_OK;
}

extern "C" int tn_pose_spaced_bins(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* origins_in,
                                   const float* directions_in, int64_t N, int32_t num_cameras, float* origins_out, float* directions_out,
                                   const float* lin_bins, const float* jitter, const float* nears, const float* fars, int32_t S, float* s_bins,
                                   float* e_bins, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(pose_adjustment && camera_indices && origins_in && directions_in && origins_out && directions_out, "tn_pose_spaced_bins: null pointer");
  TN_REQUIRE(lin_bins && nears && fars && s_bins && e_bins, "tn_pose_spaced_bins: null pointer");
  TN_REQUIRE(N >= 0 && num_cameras >= 1 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_pose_spaced_bins: bad N=%lld C=%d S=%d", (long long)N, num_cameras, S);
  const int bins_blocks = (int)std::min<int64_t>(tn_cdiv(N, 4), 4096);  // one wave per ray, 4 rays per block
  const int pose_blocks = (int)std::min<int64_t>(tn_cdiv(N, 256), (int64_t)bins_blocks);
  hipLaunchKernelGGL(k_pose_spaced_bins, dim3((unsigned)bins_blocks, 2), dim3(256), 0, tn_s(stream), pose_adjustment, frozen, camera_indices, origins_in,
                     directions_in, N, num_cameras, origins_out, directions_out, pose_blocks, lin_bins, jitter, nears, fars, S, s_bins, e_bins);
  TN_CHECK_LAUNCH("tn_pose_spaced_bins");
  return TN_OK;
}

__global__ void k_pose_bwd(const float* __restrict__ pose, const uint8_t* __restrict__ frozen, const int64_t* __restrict__ cam_idx,
                           const float* __restrict__ d_in, const float* __restrict__ g_o, const float* __restrict__ g_d, int64_t N, int C,
                           float* __restrict__ grad_pose) {
  pose_bwd_body(pose, frozen, cam_idx, d_in, g_o, g_d, N, C, grad_pose, blockIdx.x, gridDim.x);
}

// (pose_bwd_body, losses_finish_body, pose_finish_body: tn_pose_finish.h)
__global__ void __launch_bounds__(256) k_losses_finish(const float* __restrict__ lines, float* __restrict__ losses, const float* __restrict__ pose,
                                                       int C, float trans_pen, float rot_pen, float scale, float* __restrict__ reg_out,
                                                       float* __restrict__ grad_pose) {
  losses_finish_body<false>(lines, losses, pose, C, trans_pen, rot_pen, scale, reg_out, grad_pose);
}
// the end of a training iteration's backward in one launch: pose gradient from d origins / d directions (blocks 0 .. n-1) and, in the last
// block, the loss sums + the camera regulariser of the same pose tensor
__global__ void __launch_bounds__(256) k_pose_bwd_finish(const float* __restrict__ pose, const uint8_t* __restrict__ frozen,
                                                         const int64_t* __restrict__ cam_idx, const float* __restrict__ d_in,
                                                         const float* __restrict__ g_o, const float* __restrict__ g_d, int64_t N, int C,
                                                         float* __restrict__ grad_pose, const float* __restrict__ lines, float* __restrict__ losses,
                                                         float trans_pen, float rot_pen, float scale, float* __restrict__ reg_out) {
  if (blockIdx.x + 1 == gridDim.x) losses_finish_body<true>(lines, losses, pose, C, trans_pen, rot_pen, scale, reg_out, grad_pose);
  else pose_bwd_body(pose, frozen, cam_idx, d_in, g_o, g_d, N, C, grad_pose, blockIdx.x, gridDim.x - 1);
}

extern "C" int tn_pose_apply_bwd(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                                 const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose,
                                 tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(pose_adjustment && camera_indices && directions_in && d_origins && d_directions && grad_pose, "tn_pose_apply_bwd: null pointer");
  TN_REQUIRE(N >= 0 && num_cameras >= 1, "tn_pose_apply_bwd: bad N=%lld C=%d", (long long)N, num_cameras);
  if (N == 0) return TN_OK;
  hipLaunchKernelGGL(k_pose_bwd, dim3((unsigned)std::min<int64_t>(tn_cdiv(N, 256), 1024)), dim3(256), 0, tn_s(stream), pose_adjustment, frozen,
                     camera_indices, directions_in, d_origins, d_directions, N, num_cameras, grad_pose);
  TN_CHECK_LAUNCH("tn_pose_apply_bwd");
  return TN_OK;
}

extern "C" int tn_losses_finish(const float* loss_lines, float* losses16, const float* pose_adjustment, int32_t num_cameras, float trans_pen,
                                float rot_pen, float scale, float* reg_out, float* grad_pose, tn_stream_t stream) {
  TN_REQUIRE(loss_lines && losses16, "tn_losses_finish: null pointer");
  TN_REQUIRE(pose_adjustment == nullptr || (num_cameras >= 1 && reg_out != nullptr), "tn_losses_finish: bad camera regulariser arguments");
  hipLaunchKernelGGL(k_losses_finish, dim3(1), dim3(256), 0, tn_s(stream), loss_lines, losses16, pose_adjustment, num_cameras, trans_pen, rot_pen, scale,
                     reg_out, grad_pose);
  TN_CHECK_LAUNCH("tn_losses_finish");
  return TN_OK;
}

extern "C" int tn_pose_bwd_finish(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                                  const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose,
                                  const float* loss_lines, float* losses16, float trans_pen, float rot_pen, float scale, float* reg_out,
                                  tn_stream_t stream) {
  TN_REQUIRE(pose_adjustment && camera_indices && directions_in && d_origins && d_directions && grad_pose && reg_out, "tn_pose_bwd_finish: null pointer");
  TN_REQUIRE((loss_lines == nullptr) == (losses16 == nullptr), "tn_pose_bwd_finish: loss_lines and losses16 go together");
  TN_REQUIRE(N >= 1 && num_cameras >= 1, "tn_pose_bwd_finish: bad N=%lld C=%d", (long long)N, num_cameras);
  const unsigned nb = (unsigned)std::min<int64_t>(tn_cdiv(N, 256), 1024);
  hipLaunchKernelGGL(k_pose_bwd_finish, dim3(nb + 1), dim3(256), 0, tn_s(stream), pose_adjustment, frozen, camera_indices, directions_in, d_origins,
                     d_directions, N, num_cameras, grad_pose, loss_lines, losses16, trans_pen, rot_pen, scale, reg_out);
  TN_CHECK_LAUNCH("tn_pose_bwd_finish");
  return TN_OK;
}

__global__ void __launch_bounds__(256) k_pose_bwd_finish_check(PoseFinishArgs a) { pose_finish_body(a, blockIdx.x); }
// arguments of tn_pose_bwd_finish_check, validated, as the kernel's argument block (a.total_blocks = its grid)
int tn_pose_finish_args(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                        const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose, const float* loss_lines,
                        float* losses16, float trans_pen, float rot_pen, float scale, float* reg_out, const float* grads, int32_t num_ranges,
                        const int64_t* offsets, const int64_t* counts, const int32_t* flag_index, int32_t num_flags, float* found_inf, int32_t pose_flag,
                        PoseFinishArgs& a) {
  TN_REQUIRE(pose_adjustment && camera_indices && directions_in && d_origins && d_directions && grad_pose && reg_out, "tn_pose_bwd_finish_check: null pointer");
  TN_REQUIRE((loss_lines == nullptr) == (losses16 == nullptr), "tn_pose_bwd_finish_check: loss_lines and losses16 go together");
  TN_REQUIRE(N >= 1 && num_cameras >= 1, "tn_pose_bwd_finish_check: bad N=%lld C=%d", (long long)N, num_cameras);
  TN_REQUIRE(found_inf && num_flags >= 1 && pose_flag >= 0 && pose_flag < num_flags, "tn_pose_bwd_finish_check: bad found_inf / pose_flag");
  TN_REQUIRE(num_ranges >= 0 && num_ranges <= TN_ADAM_MAX_RANGES && (num_ranges == 0 || (grads && offsets && counts && flag_index)),
             "tn_pose_bwd_finish_check: bad range list");
  SmallRanges sr{};
  int nb = 0;
  for (int k = 0; k < num_ranges; ++k) {
    TN_REQUIRE(offsets[k] >= 0 && counts[k] >= 0 && counts[k] <= (1 << 22) && flag_index[k] >= 0 && flag_index[k] < num_flags,
               "tn_pose_bwd_finish_check: bad range %d (the small ranges only: at most 4 M floats each)", k);
    if (counts[k] == 0) continue;
    sr.off[sr.n] = offsets[k]; sr.cnt[sr.n] = counts[k]; sr.flag[sr.n] = flag_index[k]; sr.first_block[sr.n] = nb;
    nb += (int)tn_cdiv(counts[k], 4096);
    ++sr.n;
  }
  sr.first_block[sr.n] = nb;
  const int pb = (int)std::min<int64_t>(tn_cdiv(N, 256), 1024);
  a = PoseFinishArgs{pose_adjustment, frozen, camera_indices, directions_in, d_origins, d_directions, N, num_cameras, grad_pose, loss_lines, losses16,
                     trans_pen, rot_pen, scale, reg_out, pb, grads, sr, found_inf, (int)pose_flag, pb + 1 + nb};
  return TN_OK;
}
extern "C" int tn_pose_bwd_finish_check(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                                        const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose,
                                        const float* loss_lines, float* losses16, float trans_pen, float rot_pen, float scale, float* reg_out,
                                        const float* grads, int32_t num_ranges, const int64_t* offsets, const int64_t* counts, const int32_t* flag_index,
                                        int32_t num_flags, float* found_inf, int32_t pose_flag, tn_stream_t stream) {
  PoseFinishArgs a;
  int rc = tn_pose_finish_args(pose_adjustment, frozen, camera_indices, directions_in, d_origins, d_directions, N, num_cameras, grad_pose, loss_lines, losses16,
                               trans_pen, rot_pen, scale, reg_out, grads, num_ranges, offsets, counts, flag_index, num_flags, found_inf, pose_flag, a);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pose_bwd_finish_check, dim3((unsigned)a.total_blocks), dim3(256), 0, tn_s(stream), a);
  TN_CHECK_LAUNCH("tn_pose_bwd_finish_check");
  return TN_OK;
}

// camera_opt_regularizer = (mean_c |t_c| * trans_pen + mean_c |r_c| * rot_pen) * scale   (cameras/camera_optimizers.py:189-195)
__global__ void k_camera_reg(const float* __restrict__ pose, int C, float trans_pen, float rot_pen, float scale, float* __restrict__ loss_out,
                             float* __restrict__ grad_pose) {
  float acc = 0.0f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* p = pose + c * 6;
    float nt = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    float nr = sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
    acc += (nt * trans_pen + nr * rot_pen) * scale / (float)C;
    if (grad_pose != nullptr) {
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        // torch.norm backward: x / |x| (0 at the origin)
        if (nt > 0.0f) grad_pose[c * 6 + m] += p[m] / nt * trans_pen * scale / (float)C;
        if (nr > 0.0f) grad_pose[c * 6 + 3 + m] += p[3 + m] / nr * rot_pen * scale / (float)C;
      }
    }
  }
  acc = tn_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss_out, acc);
}

extern "C" int tn_camera_reg(const float* pose_adjustment, int32_t num_cameras, float trans_pen, float rot_pen, float scale, float* loss_out,
                             float* grad_pose, tn_stream_t stream) {
  TN_REQUIRE(pose_adjustment && loss_out && num_cameras >= 1, "tn_camera_reg: bad argument");
  hipLaunchKernelGGL(k_camera_reg, dim3(1), dim3(256), 0, tn_s(stream), pose_adjustment, num_cameras, trans_pen, rot_pen, scale, loss_out, grad_pose);
  TN_CHECK_LAUNCH("tn_camera_reg");
  return TN_OK;
}

// Per-iteration metrics of ThermalNerfactoModel.get_metrics_dict (models/thermal_nerfacto.py:253-282) that are pure functions of the loss
// sums and the pose parameters, in ONE single-block launch instead of ~10 tiny tensor operations:
//   metrics[0] = psnr_rgb     = -10 log10(rgb_loss * N / #rgb rays)                        (losses[0], losses[4] of tn_pixel_losses)
//   metrics[1] = psnr_thermal = -10 log10(thermal_loss * (N / thermal_mult) / #thermal rays) (losses[1], losses[5])
//   metrics[2 + 2k], metrics[3 + 2k] = |pose_k[:, :3]|, |pose_k[:, 3:]| (Frobenius; cameras/camera_optimizers.py:197-202), k < num_poses
__global__ void k_train_metrics(const float* __restrict__ losses, int64_t N, float thermal_mult, const float* __restrict__ pose0, int C0,
                                const float* __restrict__ pose1, int C1, float* __restrict__ metrics) {
  if (threadIdx.x == 0) {
    metrics[0] = -10.0f * log10f(losses[0] * (float)N / losses[4]);
    metrics[1] = -10.0f * log10f(losses[1] * ((float)N / thermal_mult) / losses[5]);
  }
  for (int k = 0; k < 2; ++k) {
    const float* pose = k ? pose1 : pose0;
    const int C = k ? C1 : C0;
    if (pose == nullptr) continue;
    float st = 0.0f, sr = 0.0f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float* p = pose + c * 6;
      st += p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
      sr += p[3] * p[3] + p[4] * p[4] + p[5] * p[5];
    }
    st = tn_wave_sum(st); sr = tn_wave_sum(sr);
    if (threadIdx.x == 0) { metrics[2 + 2 * k] = sqrtf(st); metrics[3 + 2 * k] = sqrtf(sr); }
  }
}

extern "C" int tn_train_metrics(const float* losses, int64_t N, float thermal_mult, const float* pose0, int32_t num_cameras0, const float* pose1,
                                int32_t num_cameras1, float* metrics_out, tn_stream_t stream) {
  TN_REQUIRE(losses && metrics_out && N >= 1 && thermal_mult > 0.0f, "tn_train_metrics: bad argument");
  TN_REQUIRE((pose0 == nullptr || num_cameras0 >= 1) && (pose1 == nullptr || num_cameras1 >= 1), "tn_train_metrics: bad camera count");
  hipLaunchKernelGGL(k_train_metrics, dim3(1), dim3(64), 0, tn_s(stream), losses, N, thermal_mult, pose0, num_cameras0, pose1, num_cameras1, metrics_out);
  TN_CHECK_LAUNCH("tn_train_metrics");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ pixel losses (body: tn_pixel_loss.h)
__global__ void __launch_bounds__(256) k_pixel_losses(const float* __restrict__ pred_rgb, int rs, const float* __restrict__ pred_th, int ts,
                                                      const float* __restrict__ image, const float* __restrict__ is_thermal, int64_t N,
                                                      float thermal_mult, float tv_mult, float cross_mult, float* __restrict__ losses,
                                                      float* __restrict__ d_rgb, float* __restrict__ d_th) {
  pixel_losses_body(pred_rgb, rs, pred_th, ts, image, is_thermal, N, thermal_mult, tv_mult, cross_mult, losses, d_rgb, d_th, blockIdx.x, gridDim.x);
}

extern "C" int tn_pixel_losses(const float* pred_rgb, int32_t rgb_stride, const float* pred_thermal, int32_t thermal_stride, const float* image,
                               const float* is_thermal, int64_t N, float thermal_mult, float tv_mult, float cross_mult, float* losses_out,
                               float* d_pred_rgb, float* d_pred_thermal, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(pred_rgb && pred_thermal && image && is_thermal && losses_out, "tn_pixel_losses: null pointer");
  TN_REQUIRE(N >= 0 && N % 4 == 0, "tn_pixel_losses: N=%lld must be a multiple of 4 (2x2 patches)", (long long)N);
  TN_REQUIRE(rgb_stride >= 3 && thermal_stride >= 1, "tn_pixel_losses: bad strides");
  if (N == 0) return TN_OK;
  hipLaunchKernelGGL(k_pixel_losses, dim3((unsigned)pixel_loss_blocks(N)), dim3(256), 0, tn_s(stream), pred_rgb, rgb_stride,
                     pred_thermal, thermal_stride, image, is_thermal, N, thermal_mult, tv_mult, cross_mult, losses_out, d_pred_rgb, d_pred_thermal);
  TN_CHECK_LAUNCH("tn_pixel_losses");
  return TN_OK;
}

// loss += a * mean|x - y| is expressed by the caller through gx/gy: d_x += gx * sign(x-y)/count, d_y -= gy * sign(x-y)/count,
// loss_out += (gx_is_loss_weight) ... see tn_l1_loss in the header: loss value uses (gx + gy) * mean|x-y|.
__global__ void k_l1(const float* __restrict__ x, const float* __restrict__ y, int64_t n, float gx, float gy, float* __restrict__ loss,
                     float* __restrict__ d_x, float* __restrict__ d_y) {
  float acc = 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float df = x[i] - y[i];
    acc += fabsf(df);
    float s = sgn(df) / (float)n;
    if (d_x) d_x[i] += gx * s;
    if (d_y) d_y[i] -= gy * s;
  }
  // one atomic per BLOCK: same-address float atomics execute one after the other (~25 ns each); one per wave of a 2048-block grid was 0.2 ms
  __shared__ float sh[4];
  acc = tn_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (gx + gy) * (sh[0] + sh[1] + sh[2] + sh[3]) / (float)n);
}
extern "C" int tn_l1_loss(const float* x, const float* y, int64_t count, float gx, float gy, float* loss_out, float* d_x, float* d_y,
                          tn_stream_t stream) {
  if (count == 0) return TN_OK;
  TN_REQUIRE(x && y && loss_out && count >= 0, "tn_l1_loss: bad argument");
  if (count == 0) return TN_OK;
  hipLaunchKernelGGL(k_l1, dim3((unsigned)std::min<int64_t>(tn_cdiv(count, 256), 256)), dim3(256), 0, tn_s(stream), x, y, count, gx, gy, loss_out,
                     d_x, d_y);
  TN_CHECK_LAUNCH("tn_l1_loss");
  return TN_OK;
}

// ------------------------------------------------------------------------------------------------ Adam
// torch.optim.Adam single-tensor arithmetic (no amsgrad / weight decay):
//   m = m*b1 + g*(1-b1) ; v = v*b2 + (1-b2)*g*g ; p += -(lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__device__ __forceinline__ void adam_body(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v,
                                          int64_t n4, float* __restrict__ pt, const float* __restrict__ gt, float* __restrict__ mt,
                                          float* __restrict__ vt, int tail, float b1, float b2, float omb1, float omb2, float neg_step,
                                          float bc2_sqrt, float eps) {
#define ADAM1(P, G, M, V)                         \
  {                                               \
    float m_new_ = M * b1 + G * omb1;             \
    float v_new_ = V * b2 + (omb2 * G) * G;       \
    float den_ = sqrtf(v_new_) / bc2_sqrt + eps;  \
    P = P + neg_step * (m_new_ / den_);           \
    M = m_new_;                                   \
    V = v_new_;                                   \
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    // gradients and moments are touched once per step: stream them past the caches (non-temporal) so that the parameters -- the hash tables the
    // next forward gathers from -- are what stays in the Infinity Cache
    typedef float v4f __attribute__((ext_vector_type(4)));
    float4 pp = p[i];
    v4f gg = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g) + i);
    v4f mm4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(m) + i);
    v4f vv4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v) + i);
    ADAM1(pp.x, gg.x, mm4.x, vv4.x)
    ADAM1(pp.y, gg.y, mm4.y, vv4.y)
    ADAM1(pp.z, gg.z, mm4.z, vv4.z)
    ADAM1(pp.w, gg.w, mm4.w, vv4.w)
    p[i] = pp;
    __builtin_nontemporal_store(mm4, reinterpret_cast<v4f*>(m) + i);
    __builtin_nontemporal_store(vv4, reinterpret_cast<v4f*>(v) + i);
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) {
    int i = threadIdx.x;
    float pp = pt[i], gg = gt[i], mm = mt[i], vv = vt[i];
    ADAM1(pp, gg, mm, vv)
    pt[i] = pp; mt[i] = mm; vt[i] = vv;
  }
#undef ADAM1
}
__global__ void k_adam(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                       float* __restrict__ pt, const float* __restrict__ gt, float* __restrict__ mt, float* __restrict__ vt, int tail, float b1,
                       float b2, float omb1, float omb2, float neg_step, float bc2_sqrt, float eps) {
  adam_body(p, g, m, v, n4, pt, gt, mt, vt, tail, b1, b2, omb1, omb2, neg_step, bc2_sqrt, eps);
}
// several ranges of the same arenas (one per optimiser group: own step count and learning rate) in one launch; blockIdx.y = range
struct AdamRanges {
  int64_t off[TN_ADAM_MAX_RANGES], cnt[TN_ADAM_MAX_RANGES];
  float neg_step[TN_ADAM_MAX_RANGES], bc2_sqrt[TN_ADAM_MAX_RANGES];
};
__global__ void k_adam_ranges(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, AdamRanges r, float b1,
                              float b2, float omb1, float omb2, float eps) {
  const int k = blockIdx.y;
  const int64_t off = r.off[k], n = r.cnt[k], n4 = n / 4;
  if ((int64_t)blockIdx.x * blockDim.x >= n4 && blockIdx.x != 0) return;  // short ranges need few blocks
  adam_body(reinterpret_cast<float4*>(p + off), reinterpret_cast<const float4*>(g + off), reinterpret_cast<float4*>(m + off),
            reinterpret_cast<float4*>(v + off), n4, p + off + n4 * 4, g + off + n4 * 4, m + off + n4 * 4, v + off + n4 * 4, (int)(n - n4 * 4), b1, b2, omb1,
            omb2, r.neg_step[k], r.bc2_sqrt[k], eps);
}

extern "C" int tn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, int32_t step, double lr,
                            double beta1, double beta2, double eps, tn_stream_t stream) {
  if (count == 0) return TN_OK;
  TN_REQUIRE(params && grads && exp_avg && exp_avg_sq, "tn_adam_step: null pointer");
  TN_REQUIRE(count >= 0 && step >= 1, "tn_adam_step: bad count=%lld step=%d", (long long)count, step);
  TN_REQUIRE(((uintptr_t)params % 16 == 0) && ((uintptr_t)grads % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) && ((uintptr_t)exp_avg_sq % 16 == 0),
             "tn_adam_step: arena pointers must be 16-byte aligned");
  if (count == 0) return TN_OK;
  double bc1 = 1.0 - pow(beta1, (double)step);
  double bc2 = 1.0 - pow(beta2, (double)step);
  float neg_step = (float)(-(lr / bc1));
  float bc2_sqrt = (float)sqrt(bc2);
  int64_t n4 = count / 4;
  int tail = (int)(count - n4 * 4);
  int grid = (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(n4, 256), 256 * 16));
  hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, tn_s(stream), (float4*)params, (const float4*)grads, (float4*)exp_avg, (float4*)exp_avg_sq, n4,
                     params + n4 * 4, grads + n4 * 4, exp_avg + n4 * 4, exp_avg_sq + n4 * 4, tail, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                     (float)(1.0 - beta2), neg_step, bc2_sqrt, (float)eps);
  TN_CHECK_LAUNCH("tn_adam_step");
  return TN_OK;
}

extern "C" int tn_adam_step_ranges(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                                   const int64_t* counts, const int32_t* steps, const double* lrs, double beta1, double beta2, double eps,
                                   tn_stream_t stream) {
  if (num_ranges == 0) return TN_OK;
  TN_REQUIRE(params && grads && exp_avg && exp_avg_sq && offsets && counts && steps && lrs, "tn_adam_step_ranges: null pointer");
  TN_REQUIRE(num_ranges > 0 && num_ranges <= TN_ADAM_MAX_RANGES, "tn_adam_step_ranges: %d ranges (at most %d)", num_ranges, TN_ADAM_MAX_RANGES);
  TN_REQUIRE(((uintptr_t)params % 16 == 0) && ((uintptr_t)grads % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) && ((uintptr_t)exp_avg_sq % 16 == 0),
             "tn_adam_step_ranges: arena pointers must be 16-byte aligned");
  AdamRanges r{};
  int64_t max_n4 = 0;
  int n = 0;
  for (int k = 0; k < num_ranges; ++k) {
    TN_REQUIRE(offsets[k] >= 0 && offsets[k] % 4 == 0 && counts[k] >= 0 && steps[k] >= 1, "tn_adam_step_ranges: bad range %d (offset %lld count %lld step %d)",
               k, (long long)offsets[k], (long long)counts[k], steps[k]);
    if (counts[k] == 0) continue;
    double bc1 = 1.0 - pow(beta1, (double)steps[k]);
    double bc2 = 1.0 - pow(beta2, (double)steps[k]);
    r.off[n] = offsets[k];
    r.cnt[n] = counts[k];
    r.neg_step[n] = (float)(-(lrs[k] / bc1));
    r.bc2_sqrt[n] = (float)sqrt(bc2);
    max_n4 = std::max<int64_t>(max_n4, counts[k] / 4);
    ++n;
  }
  if (n == 0) return TN_OK;
  int grid = (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(max_n4, 256), 256 * 16));
  hipLaunchKernelGGL(k_adam_ranges, dim3(grid, n), dim3(256), 0, tn_s(stream), params, grads, exp_avg, exp_avg_sq, r, (float)beta1, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps);
  TN_CHECK_LAUNCH("tn_adam_step_ranges");
  return TN_OK;
}

// ---- GradScaler semantics on the device (torch/amp/grad_scaler.py as engine/trainer.py:470-495 drives it) ------------------------------
// found_inf: set to 1 if any gradient is non-finite (never cleared here: the caller zero-fills it once per step and may check several slices)
__global__ void __launch_bounds__(256) k_grad_nonfinite(const float4* __restrict__ g, int64_t n4, const float* __restrict__ gt, int tail,
                                                         float* __restrict__ found_inf) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const v4f v = reinterpret_cast<const v4f*>(g)[i];
    // x - x is 0 for finite x, NaN for inf / NaN
    const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
    bad = bad || (t != 0.0f);
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) { const float x = gt[threadIdx.x]; bad = bad || ((x - x) != 0.0f); }
  if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.0f;  // plain store of the same value from any number of waves
}
extern "C" int tn_grad_nonfinite(const float* grads, int64_t count, float* found_inf, tn_stream_t stream) {
  if (count == 0) return TN_OK;
  TN_REQUIRE(grads && found_inf && count > 0, "tn_grad_nonfinite: bad argument");
  TN_REQUIRE(((uintptr_t)grads % 16) == 0, "tn_grad_nonfinite: grads must be 16-byte aligned");
  const int64_t n4 = count / 4;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(n4, 256 * 4), 256 * 8));
  hipLaunchKernelGGL(k_grad_nonfinite, dim3(grid), dim3(256), 0, tn_s(stream), reinterpret_cast<const float4*>(grads), n4, grads + n4 * 4, (int)(count - n4 * 4),
                     found_inf);
  TN_CHECK_LAUNCH("tn_grad_nonfinite");
  return TN_OK;
}
// several ranges of one gradient arena in ONE launch (blockIdx.y = range): range k raises found_inf[flag_index[k]]
struct NonfiniteRanges { int64_t off[TN_ADAM_MAX_RANGES], cnt[TN_ADAM_MAX_RANGES]; int32_t flag[TN_ADAM_MAX_RANGES]; };
__global__ void __launch_bounds__(256) k_grad_nonfinite_ranges(const float* __restrict__ g, NonfiniteRanges r, float* __restrict__ found_inf) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int k = blockIdx.y;
  const int64_t n = r.cnt[k], n4 = n / 4;
  if ((int64_t)blockIdx.x * blockDim.x >= n4 && blockIdx.x != 0) return;
  const float* base = g + r.off[k];
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const v4f v = reinterpret_cast<const v4f*>(base)[i];
    const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
    bad = bad || (t != 0.0f);
  }
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < n - n4 * 4) { const float x = base[n4 * 4 + threadIdx.x]; bad = bad || ((x - x) != 0.0f); }
  if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[r.flag[k]] = 1.0f;
}
extern "C" int tn_grad_nonfinite_ranges(const float* grads, int32_t num_ranges, const int64_t* offsets, const int64_t* counts, const int32_t* flag_index,
                                        int32_t num_flags, float* found_inf, tn_stream_t stream) {
  if (num_ranges == 0) return TN_OK;
  TN_REQUIRE(grads && offsets && counts && found_inf && num_ranges > 0 && num_ranges <= TN_ADAM_MAX_RANGES && num_flags >= 1,
             "tn_grad_nonfinite_ranges: bad argument");
  TN_REQUIRE(((uintptr_t)grads % 16) == 0, "tn_grad_nonfinite_ranges: grads must be 16-byte aligned");
  NonfiniteRanges r{};
  int64_t max_n4 = 0;
  int n = 0;
  for (int k = 0; k < num_ranges; ++k) {
    const int fl = flag_index ? flag_index[k] : 0;
    TN_REQUIRE(offsets[k] >= 0 && offsets[k] % 4 == 0 && counts[k] >= 0 && fl >= 0 && fl < num_flags, "tn_grad_nonfinite_ranges: bad range %d", k);
    if (counts[k] == 0) continue;
    r.off[n] = offsets[k]; r.cnt[n] = counts[k]; r.flag[n] = fl;
    max_n4 = std::max<int64_t>(max_n4, counts[k] / 4);
    ++n;
  }
  if (n == 0) return TN_OK;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(max_n4, 256 * 4), 256 * 8));
  hipLaunchKernelGGL(k_grad_nonfinite_ranges, dim3(grid, n), dim3(256), 0, tn_s(stream), grads, r, found_inf);
  TN_CHECK_LAUNCH("tn_grad_nonfinite_ranges");
  return TN_OK;
}

// Adam with the skip / unscale decision ON THE DEVICE: no host synchronisation between backward and optimiser step.  GradScaler decides per
// OPTIMISER (torch/amp/grad_scaler.py: found_inf_per_device of that optimiser's gradients), i.e. per parameter group here:
//   found_inf[flag[k]] != 0 -> range k is not touched (parameters and both moments stay bit-identical); if `count_skip`, skipped[flag[k]] += 1
//                              (once per flag and launch: a group whose parameters form several ranges is one skipped step)
//   inv_scale               -> gradients are multiplied by *inv_scale as they are read (GradScaler.step on not-yet-unscaled gradients)
//   skipped[flag[k]]        -> steps of that group skipped so far: the bias corrections use (host step count - skipped), as torch's fused Adam
//                              keeps its step tensors (torch/optim/adam.py _fused_adam: step -= found_inf)
//   skipped[lag_index]      -> iterations in which ANY group found an inf (the scale dropped, the trainer did not step the schedulers): the LR
//                              schedule is evaluated at sched_step - that; tn_grad_scaler_update maintains it
struct AdamRangesAmp {
  int64_t off[TN_ADAM_MAX_RANGES], cnt[TN_ADAM_MAX_RANGES];
  int32_t step[TN_ADAM_MAX_RANGES], flag[TN_ADAM_MAX_RANGES];
  double lr[TN_ADAM_MAX_RANGES];        // the learning rate, or lr_init of the schedule below
  double lr_final[TN_ADAM_MAX_RANGES];  // ExponentialDecayScheduler (engine/schedulers.py:109-141) evaluated on the device when max_steps > 0:
  int32_t max_steps[TN_ADAM_MAX_RANGES];  //   lr = exp(lerp(log lr_init, log lr_final, clip((sched_step - lag) / max_steps, 0, 1)))
  int32_t sched_step;                     // scheduler steps on the host's count
  int32_t num_flags, lag_index;           // entries of found_inf; index of the schedule lag in skipped (-1: none)
};
// GradScaler.update() by the LAST block of the Adam launch to finish (tn_adam_step_ranges_amp_update): every block has read found_inf and the
// schedule lag by the time it counts itself done, so the block that sees the full count may rewrite them -- one launch (4.6 us on the serial
// chain of a step) less.  done: one zeroed uint32 on the device, left zero again.
struct ScalerUpdate {
  float* scale;             // NULL: no fused update
  int32_t* growth_tracker;
  uint32_t* done;
  float growth_factor, backoff_factor;
  int32_t growth_interval;
};
__device__ __forceinline__ void scaler_update_body(float* scale, int32_t* growth_tracker, float* found_inf, int num_flags, int32_t* lag, float growth_factor,
                                                   float backoff_factor, int growth_interval, int clear) {
  bool any = false;
  for (int i = 0; i < num_flags; ++i) {
    any = any || (found_inf[i] != 0.0f);
    if (clear) found_inf[i] = 0.0f;  // ready for the next iteration: no separate zero-fill launch
  }
  if (any) {
    *scale = *scale * backoff_factor;
    *growth_tracker = 0;
    if (lag != nullptr) *lag += 1;
  } else {
    const int successful = *growth_tracker + 1;
    if (successful == growth_interval) {
      const float ns = *scale * growth_factor;
      if ((ns - ns) == 0.0f) *scale = ns;  // finite
      *growth_tracker = 0;
    } else {
      *growth_tracker = successful;
    }
  }
}
// Counting the blocks in: atomics into ONE 64-byte line execute one after the other (~25 ns each), and the launch has up to 12 k blocks that
// all finish at about the same time -- a single counter made the launch 4.5 x longer (0.3 ms of serialised atomics).  So the blocks count
// themselves on 64 counters in 64 different lines (block b on line b mod 64), and the block that completes a line counts the LINE on a 65th.
#define TN_DONE_WORDS (65 * 16)  // uint32 words of the done-counter buffer (all zero before the launch, all zero after it)
// No fences: a device-scope release fence on gfx950 writes the XCD's whole L2 back (the launch has 300 MB of dirty lines: with one
// __threadfence() per block it took 535 us instead of 67).  None is needed: what must be ordered is every block's READS of found_inf / the lag
// before the updater's writes, and a block counts itself in only after its threads have USED those values (the barrier below); counters are
// relaxed device-scope atomics, their resets and the update itself are consumed by later launches only.
__device__ __forceinline__ void adam_block_done(const ScalerUpdate& su, float* found_inf, int num_flags, int32_t* skipped, int lag_index) {
  if (su.scale == nullptr) return;
  __syncthreads();  // every thread of the block has read what it needs of found_inf / skipped
  if (threadIdx.x == 0) {
    const uint32_t total = gridDim.x * gridDim.y, lb = blockIdx.y * gridDim.x + blockIdx.x, line = lb & 63u;
    const uint32_t expect = total / 64u + ((line < (total & 63u)) ? 1u : 0u);
    if (__hip_atomic_fetch_add(su.done + line * 16u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == expect) {
      __hip_atomic_store(su.done + line * 16u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t lines = total < 64u ? total : 64u;
      if (__hip_atomic_fetch_add(su.done + 64u * 16u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == lines) {
        __hip_atomic_store(su.done + 64u * 16u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scaler_update_body(su.scale, su.growth_tracker, found_inf, num_flags, (skipped != nullptr && lag_index >= 0) ? skipped + lag_index : nullptr,
                           su.growth_factor, su.backoff_factor, su.growth_interval, 1);
      }
    }
  }
}
struct SampleCoWork { SamplePixelsArgs a; RaygenArgs g; int blocks; };
#include "tn_next_sampling.h"
// block `bid` of the `nblk` blocks that walk range k of the launch (grid-stride)
__device__ __forceinline__ void adam_range_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                const AdamRangesAmp& r, double beta1, double beta2, float eps, const float* __restrict__ inv_scale,
                                                float* __restrict__ found_inf, int32_t* __restrict__ skipped, int count_skip, int zero_g, int k, int bid,
                                                int nblk) {
  const int fl = r.flag[k];
  float* gz = const_cast<float*>(g);  // zero_g: the gradients are consumed (set to zero behind the read): no zero-fill launch before the next backward
  // schedule lag = iterations so far in which the scale dropped; maintained by tn_grad_scaler_update AFTER this launch (stream order)
  const int lag = (skipped != nullptr && r.lag_index >= 0) ? skipped[r.lag_index] : 0;
  const int64_t off = r.off[k], n = r.cnt[k], n4 = n / 4;
  const int64_t stride = (int64_t)nblk * blockDim.x;
  if (found_inf != nullptr && found_inf[fl] != 0.0f) {  // uniform over the range's blocks
    if (count_skip && skipped != nullptr && bid == 0 && threadIdx.x == 0) {
      bool first = true;  // ONE count per group and launch, however many ranges of the launch carry the group's flag
      for (int j = 0; j < k; ++j) first = first && r.flag[j] != fl;
      if (first) atomicAdd(&skipped[fl], 1);
    }
    if (zero_g) {  // the skipped step still consumes its (non-finite) gradients
      for (int64_t i = bid * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4*>(gz + off)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bid == 0 && (int64_t)threadIdx.x < n - n4 * 4) gz[off + n4 * 4 + threadIdx.x] = 0.0f;
    }
    return;
  }
  if ((int64_t)bid * blockDim.x >= n4 && bid != 0) return;
  __shared__ float s_ns, s_bc;
  __syncthreads();  // (a block that walks several ranges: the previous range's readers are done with s_ns / s_bc)
  if (threadIdx.x == 0) {
    const int sk = skipped ? skipped[fl] : 0;
    const int eff = r.step[k] - sk;
    const double bc1 = 1.0 - pow(beta1, (double)(eff < 1 ? 1 : eff)), bc2 = 1.0 - pow(beta2, (double)(eff < 1 ? 1 : eff));
    double lr = r.lr[k];
    if (r.max_steps[k] > 0) {
      double t = (double)(r.sched_step - lag) / (double)r.max_steps[k];
      t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
      lr = exp(log(r.lr[k]) * (1.0 - t) + log(r.lr_final[k]) * t);
    }
    s_ns = (float)(-(lr / bc1));
    s_bc = (float)sqrt(bc2);
  }
  __syncthreads();
  const float neg_step = s_ns, bc2_sqrt = s_bc, b1 = (float)beta1, b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  const float is = inv_scale ? *inv_scale : 1.0f;
  float4* p4 = reinterpret_cast<float4*>(p + off);
  const float4* g4 = reinterpret_cast<const float4*>(g + off);
  float4* m4 = reinterpret_cast<float4*>(m + off);
  float4* v4 = reinterpret_cast<float4*>(v + off);
#define ADAM1(P, G, M, V)                         \
  {                                               \
    float g_ = inv_scale ? (G) * is : (G);        \
    float m_new_ = M * b1 + g_ * omb1;            \
    float v_new_ = V * b2 + (omb2 * g_) * g_;     \
    float den_ = sqrtf(v_new_) / bc2_sqrt + eps;  \
    P = P + neg_step * (m_new_ / den_);           \
    M = m_new_;                                   \
    V = v_new_;                                   \
  }
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  // Entries that have never received a gradient (g, m and v all +0: on the coarse levels of a hash table most slots -- level 0 uses 4913 of
  // 2^19) stay exactly as they are under Adam's arithmetic (m' = v' = 0, p' = p + step * 0 / eps = p): nothing to read further, nothing to write.
#define ADAM4(I, GG, MM, VV)                                                                                           \
  {                                                                                                                    \
    const v4u zb = __builtin_bit_cast(v4u, GG) | __builtin_bit_cast(v4u, MM) | __builtin_bit_cast(v4u, VV);          \
    if ((zb.x | zb.y | zb.z | zb.w) != 0u) {                                                                           \
      float4 pp = p4[I];                                                                                               \
      ADAM1(pp.x, GG.x, MM.x, VV.x)                                                                                    \
      ADAM1(pp.y, GG.y, MM.y, VV.y)                                                                                    \
      ADAM1(pp.z, GG.z, MM.z, VV.z)                                                                                    \
      ADAM1(pp.w, GG.w, MM.w, VV.w)                                                                                    \
      p4[I] = pp;                                                                                                      \
      __builtin_nontemporal_store(MM, reinterpret_cast<v4f*>(m4) + (I));                                               \
      __builtin_nontemporal_store(VV, reinterpret_cast<v4f*>(v4) + (I));                                               \
      if (zero_g) { const v4f z = {0.f, 0.f, 0.f, 0.f}; __builtin_nontemporal_store(z, reinterpret_cast<v4f*>(gz + off) + (I)); } \
    }                                                                                                                  \
  }
  int64_t i0 = bid * (int64_t)blockDim.x + threadIdx.x;
  for (; i0 < n4; i0 += stride) {
    v4f gg = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g4) + i0);
    v4f mm4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(m4) + i0);
    v4f vv4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v4) + i0);
    ADAM4(i0, gg, mm4, vv4)
  }
#undef ADAM4
  const int tail = (int)(n - n4 * 4);
  if (bid == 0 && (int)threadIdx.x < tail) {
    const int64_t i = off + n4 * 4 + threadIdx.x;
    float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    ADAM1(pp, gg, mm, vv)
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (zero_g) gz[i] = 0.0f;
  }
#undef ADAM1
}
__global__ void k_adam_ranges_amp(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, AdamRangesAmp r,
                                  double beta1, double beta2, float eps, const float* __restrict__ inv_scale, float* __restrict__ found_inf,
                                  int32_t* __restrict__ skipped, int count_skip, int zero_g, ScalerUpdate su, SampleCoWork cw) {
  int k = blockIdx.y;
  if (cw.blocks > 0) {
    // co-work row (row 0, dispatched first): the NEXT iteration's pixel sampling + ray generation -- a short latency-bound pass beside this
    // HBM-bound one, instead of 7 us at the head of the next iteration.  Its blocks count themselves in like every block of the launch.
    if (k == 0) {
      if (blockIdx.x < (unsigned)cw.blocks) sample_rays_body(cw.a, cw.g, blockIdx.x, (unsigned)cw.blocks);
      adam_block_done(su, found_inf, r.num_flags, skipped, r.lag_index);
      return;
    }
    --k;
  }
  adam_range_body(p, g, m, v, r, beta1, beta2, eps, inv_scale, found_inf, skipped, count_skip, zero_g, k, (int)blockIdx.x, (int)gridDim.x);
  adam_block_done(su, found_inf, r.num_flags, skipped, r.lag_index);
}
// The optimiser launch that carries the NEXT iteration's sampling front (tn_next_sampling.h: pose correction + both proposal levels, one wave per
// ray) in a co-work row in front of its range rows.  A kernel of its own: the chain's 127 registers and 18 KB of LDS per block would otherwise be
// every Adam launch's.
// What the launch gains (profiles/r06_next_sampling.md): the chain is bound by vector-instruction issue (~12 k wave-instructions per ray; 80 us for
// 4096 rays when it has the chip to itself) and at 127 registers its 1024 blocks are exactly the chip's 4 waves per SIMD -- so the range rows
// start as chain blocks retire and the launch takes 141 us for 80 + 78, not the ~85 us a perfect overlap would.  The launch still wins over the five
// in-line launches it replaces (their gaps, the proposal update's larger first launch); two layouts that make room for both kinds at once -- 5 or 6
// waves per SIMD with persistent Adam blocks (1 or 2 per CU, 4 x the loads in flight per lane) interleaved 1 : 4 / 2 : 4 with the chain blocks --
// were built and measured SLOWER (177 / 196 us): the chain spills 26 / 47 registers there and HBM wants ~100 KB in flight per CU, which one or two
// Adam waves per SIMD do not provide.
// FUSED (TN_NEXT_SAMPLING=4, an experiment): the chain's waves step the first fa.batches batches of the launch's one range themselves, between
// their stages (tn_next_sampling.h: FusedAdam); the range row takes the rest.
template <bool FUSED>
__global__ void __launch_bounds__(256, 4) k_adam_ranges_amp_next(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                                 AdamRangesAmp r, double beta1, double beta2, float eps, const float* __restrict__ inv_scale,
                                                                 float* __restrict__ found_inf, int32_t* __restrict__ skipped, int count_skip, int zero_g,
                                                                 ScalerUpdate su, NextSamplingArgs ns, FusedAdam fa) {
  __shared__ __attribute__((aligned(16))) float ns_lds[NS_LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float fa_stage[FUSED ? 4 * FA_STAGE_FLOATS : 4];
  __shared__ float fa_scal[4];
  if (blockIdx.y == 0) {
    if (blockIdx.x < (unsigned)ns.blocks) next_sampling_body<FUSED>(ns, blockIdx.x, (unsigned)ns.blocks, ns_lds, &fa, fa_stage, fa_scal);
  } else {
    adam_range_body(p, g, m, v, r, beta1, beta2, eps, inv_scale, found_inf, skipped, count_skip, zero_g, (int)blockIdx.y - 1, (int)blockIdx.x, (int)gridDim.x);
  }
  adam_block_done(su, found_inf, r.num_flags, skipped, r.lag_index);
}
// (the chain alone, for register accounting and A/B timing: TN_NEXT_SAMPLING=2 launches it behind the optimiser launch instead of inside it)
__global__ void __launch_bounds__(256, 4) k_next_sampling(NextSamplingArgs ns) {
  __shared__ __attribute__((aligned(16))) float ns_lds[NS_LDS_FLOATS];
  next_sampling_body(ns, blockIdx.x, gridDim.x, ns_lds);
}
static int adam_ranges_amp_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                               const int64_t* counts, const int32_t* steps, const double* lrs, const double* lr_finals,
                               const int32_t* sched_max_steps, int32_t sched_step, double beta1, double beta2, double eps,
                               const float* inv_scale, const float* found_inf, const int32_t* flag_index, int32_t num_flags, int32_t* skipped,
                               int32_t lag_index, int32_t count_skip, int32_t zero_grads, ScalerUpdate su, tn_stream_t stream,
                               const TnSampleRays* next = nullptr, bool* next_taken = nullptr, const TnNextSamplingHost* chain = nullptr) {
  if (next_taken) *next_taken = false;
  if (num_ranges == 0) return TN_OK;
  TN_REQUIRE(params && grads && exp_avg && exp_avg_sq && offsets && counts && steps && lrs, "tn_adam_step_ranges_amp: null pointer");
  TN_REQUIRE(num_ranges > 0 && num_ranges <= TN_ADAM_MAX_RANGES, "tn_adam_step_ranges_amp: %d ranges (at most %d)", num_ranges, TN_ADAM_MAX_RANGES);
  TN_REQUIRE(((uintptr_t)params % 16 == 0) && ((uintptr_t)grads % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) && ((uintptr_t)exp_avg_sq % 16 == 0),
             "tn_adam_step_ranges_amp: arena pointers must be 16-byte aligned");
  TN_REQUIRE(num_flags >= 1 && num_flags <= 64 && lag_index >= -1 && lag_index <= 64, "tn_adam_step_ranges_amp: bad num_flags / lag_index");
  AdamRangesAmp r{};
  int64_t max_n4 = 0;
  int n = 0;
  for (int k = 0; k < num_ranges; ++k) {
    TN_REQUIRE(offsets[k] >= 0 && offsets[k] % 4 == 0 && counts[k] >= 0 && steps[k] >= 1, "tn_adam_step_ranges_amp: bad range %d (offset %lld count %lld step %d)",
               k, (long long)offsets[k], (long long)counts[k], steps[k]);
    const int fl = flag_index ? flag_index[k] : 0;
    TN_REQUIRE(fl >= 0 && fl < num_flags, "tn_adam_step_ranges_amp: flag index %d of range %d outside [0, %d)", fl, k, num_flags);
    if (counts[k] == 0) continue;
    r.off[n] = offsets[k]; r.cnt[n] = counts[k]; r.step[n] = steps[k]; r.lr[n] = lrs[k]; r.flag[n] = fl;
    r.lr_final[n] = lr_finals ? lr_finals[k] : lrs[k];
    r.max_steps[n] = (lr_finals && sched_max_steps) ? sched_max_steps[k] : 0;
    max_n4 = std::max<int64_t>(max_n4, counts[k] / 4);
    ++n;
  }
  if (n == 0) return TN_OK;
  r.sched_step = sched_step;
  r.num_flags = num_flags;
  r.lag_index = lag_index;
  int grid = (int)std::max<int64_t>(1, std::min<int64_t>(tn_cdiv(max_n4, 256), 256 * 16));
  SampleCoWork cw{};
  if (next != nullptr && next->num_rays > 0) {
    int rc = sample_rays_build("tn_train_step(next_sample)", next, cw.a, cw.g);
    if (rc) return rc;
    // 4 lanes per ray.  The row has `grid` blocks: a launch over small ranges only (the pose corrections alone) is widened for its co-work row --
    // the range rows' surplus blocks leave at once
    grid = std::max(grid, (int)std::min<int64_t>(tn_cdiv(next->num_rays * 4, 256), 256));
    cw.blocks = (int)std::min<int64_t>(tn_cdiv(next->num_rays * 4, 256), grid);
    if (next_taken) *next_taken = true;
  }
  if (chain != nullptr) {
    // the next iteration's sampling front for the batch chain->rays_* hold (sampled by an earlier launch): one wave per ray, 4 rays per block and trip
    TN_REQUIRE(next == nullptr, "tn_train_step(next_sampling): the launch that carries the chain does not sample the batch");
    TN_REQUIRE(chain->N > 0 && chain->N % 4 == 0 && chain->N < (1ll << 30), "tn_train_step(next_sampling): %lld rays (a multiple of 4 is needed)", (long long)chain->N);
    TN_REQUIRE(tn_next_sampling_supported(chain->S0, chain->S1, chain->S2), "tn_train_step(next_sampling): unsupported sample counts (%d, %d, %d)", chain->S0,
               chain->S1, chain->S2);
    TN_REQUIRE(chain->prop0 && chain->prop1 && chain->pose && chain->nears && chain->fars && chain->lin0 && chain->lin1 && chain->lin2 && chain->out &&
                   chain->rays_o && chain->rays_d && chain->cam && chain->num_cameras >= 1,
               "tn_train_step(next_sampling): null pointer");
    TN_REQUIRE(chain->prop0->grid.num_levels == PL && chain->prop1->grid.num_levels == PL && chain->prop0->grid.log2_hashmap_size >= 1 &&
                   chain->prop0->grid.log2_hashmap_size <= 24 && chain->prop1->grid.log2_hashmap_size >= 1 && chain->prop1->grid.log2_hashmap_size <= 24,
               "tn_train_step(next_sampling): proposal grids are built for %d levels", PL);
    NextSamplingArgs ns{};
    auto prop = [](const TnPropNet* q) {
      NsProp o{};
      o.table = reinterpret_cast<const float2*>(q->grid.table);
      o.tsize = 1u << q->grid.log2_hashmap_size; o.mask = o.tsize - 1u;
      for (int l = 0; l < PL; ++l) o.res[l] = q->grid.res[l];
      o.w0 = q->w0; o.b0 = q->b0; o.w1 = q->w1; o.b1 = q->b1;
      return o;
    };
    TN_REQUIRE(chain->prop0->grid.table && chain->prop0->w0 && chain->prop0->b0 && chain->prop0->w1 && chain->prop0->b1 && chain->prop1->grid.table &&
                   chain->prop1->w0 && chain->prop1->b0 && chain->prop1->w1 && chain->prop1->b1, "tn_train_step(next_sampling): null parameter pointer");
    ns.p0 = prop(chain->prop0); ns.p1 = prop(chain->prop1);
    ns.pose = chain->pose; ns.frozen = chain->frozen; ns.num_cameras = chain->num_cameras;
    ns.rays_o = chain->rays_o; ns.rays_d = chain->rays_d; ns.cam = chain->cam;
    ns.nears = chain->nears; ns.fars = chain->fars;
    ns.jit0 = chain->jit0; ns.jit1 = chain->jit1; ns.jit2 = chain->jit2;
    ns.lin0 = chain->lin0; ns.lin1 = chain->lin1; ns.lin2 = chain->lin2;
    ns.anneal = chain->anneal;
    ns.S0 = chain->S0; ns.S1 = chain->S1; ns.S2 = chain->S2; ns.N = (int)chain->N;
    ns.out = chain->out;
    for (int k = 0; k < NS_SLOTS; ++k) {
      TN_REQUIRE(chain->off[k] >= 0 && chain->off[k] < (1ll << 32), "tn_train_step(next_sampling): forward buffer region %d out of range", k);
      ns.off[k] = (uint32_t)chain->off[k];
    }
    ns.save_enc = chain->save_enc;
    grid = std::max(grid, (int)std::min<int64_t>(chain->N / 4, 1024));  // (a launch over small ranges only is widened for its co-work row)
    ns.blocks = (int)std::min<int64_t>(chain->N / 4, grid);
    const char* mode = getenv("TN_NEXT_SAMPLING");
    if (mode && mode[0] == '3') {  // A/B timing: the chain on a companion stream beside the optimiser launch; tn_train_step joins it before the next field forward
      hipStream_t side = tn_fork_n(tn_s(stream), 3);
      hipLaunchKernelGGL(k_next_sampling, dim3(ns.blocks), dim3(256), 0, side ? side : tn_s(stream), ns);
      TN_CHECK_LAUNCH("tn_next_sampling");
      SampleCoWork none{};
      hipLaunchKernelGGL(k_adam_ranges_amp, dim3(grid, n), dim3(256), 0, tn_s(stream), params, grads, exp_avg, exp_avg_sq, r, beta1, beta2, (float)eps, inv_scale,
                         const_cast<float*>(found_inf), skipped, (int)count_skip, (int)zero_grads, su, none);
      TN_CHECK_LAUNCH("tn_adam_step_ranges_amp");
      return TN_OK;
    }
    if (mode && mode[0] == '2') {  // A/B timing: the chain as a launch of its own behind the optimiser launch (same results)
      SampleCoWork none{};
      hipLaunchKernelGGL(k_adam_ranges_amp, dim3(grid, n), dim3(256), 0, tn_s(stream), params, grads, exp_avg, exp_avg_sq, r, beta1, beta2, (float)eps, inv_scale,
                         const_cast<float*>(found_inf), skipped, (int)count_skip, (int)zero_grads, su, none);
      TN_CHECK_LAUNCH("tn_adam_step_ranges_amp");
      hipLaunchKernelGGL(k_next_sampling, dim3(ns.blocks), dim3(256), 0, tn_s(stream), ns);
      TN_CHECK_LAUNCH("tn_next_sampling");
      return TN_OK;
    }
    FusedAdam fa{};
    if (mode && mode[0] == '4' && n == 1 && inv_scale == nullptr) {
      // the chain's waves step the head of the (one) range: as many 256-float batches as their service points can take; the range row the rest
      const int64_t nb = r.cnt[0] / 256, W = (int64_t)ns.blocks * 4, trips = tn_cdiv(chain->N, W);
      const char* fs = getenv("TN_FUSED_SITES");  // (tuning aid: batches per ray the host hands to the chain's waves; default = what their stages take)
      const int64_t per_ray = fs ? std::max(1, atoi(fs)) : FA_SITES_PER_RAY;
      const int64_t take = std::min<int64_t>(nb - 1, per_ray * trips * W);
      if (take >= W) {
        fa.p = params + r.off[0]; fa.g = grads + r.off[0]; fa.m = exp_avg + r.off[0]; fa.v = exp_avg_sq + r.off[0];
        fa.batches = (uint32_t)take;
        fa.flag = r.flag[0]; fa.step = r.step[0]; fa.max_steps = r.max_steps[0]; fa.sched_step = r.sched_step; fa.lag_index = r.lag_index;
        fa.zero_g = (int)zero_grads;
        fa.beta1 = beta1; fa.beta2 = beta2; fa.lr = r.lr[0]; fa.lr_final = r.lr_final[0]; fa.eps = (float)eps;
        fa.found_inf = found_inf; fa.skipped = skipped;
        r.off[0] += take * 256; r.cnt[0] -= take * 256;
      }
    }
    if (fa.batches > 0)
      hipLaunchKernelGGL(k_adam_ranges_amp_next<true>, dim3(grid, n + 1), dim3(256), 0, tn_s(stream), params, grads, exp_avg, exp_avg_sq, r, beta1, beta2, (float)eps,
                         inv_scale, const_cast<float*>(found_inf), skipped, (int)count_skip, (int)zero_grads, su, ns, fa);
    else
      hipLaunchKernelGGL(k_adam_ranges_amp_next<false>, dim3(grid, n + 1), dim3(256), 0, tn_s(stream), params, grads, exp_avg, exp_avg_sq, r, beta1, beta2, (float)eps,
                         inv_scale, const_cast<float*>(found_inf), skipped, (int)count_skip, (int)zero_grads, su, ns, fa);
    TN_CHECK_LAUNCH("tn_adam_step_ranges_amp(next_sampling)");
    return TN_OK;
  }
  hipLaunchKernelGGL(k_adam_ranges_amp, dim3(grid, n + (cw.blocks > 0 ? 1 : 0)), dim3(256), 0, tn_s(stream), params, grads, exp_avg, exp_avg_sq, r, beta1, beta2,
                     (float)eps, inv_scale, const_cast<float*>(found_inf), skipped, (int)count_skip, (int)zero_grads, su, cw);
  TN_CHECK_LAUNCH("tn_adam_step_ranges_amp");
  return TN_OK;
}
extern "C" int tn_adam_step_ranges_amp(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                                       const int64_t* counts, const int32_t* steps, const double* lrs, const double* lr_finals,
                                       const int32_t* sched_max_steps, int32_t sched_step, double beta1, double beta2, double eps,
                                       const float* inv_scale, const float* found_inf, const int32_t* flag_index, int32_t num_flags, int32_t* skipped,
                                       int32_t lag_index, int32_t count_skip, int32_t zero_grads, tn_stream_t stream) {
  return adam_ranges_amp_impl(params, grads, exp_avg, exp_avg_sq, num_ranges, offsets, counts, steps, lrs, lr_finals, sched_max_steps, sched_step, beta1, beta2,
                              eps, inv_scale, found_inf, flag_index, num_flags, skipped, lag_index, count_skip, zero_grads, ScalerUpdate{}, stream);
}
extern "C" int tn_adam_step_ranges_amp_update(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                                              const int64_t* counts, const int32_t* steps, const double* lrs, const double* lr_finals,
                                              const int32_t* sched_max_steps, int32_t sched_step, double beta1, double beta2, double eps,
                                              const float* inv_scale, float* found_inf, const int32_t* flag_index, int32_t num_flags, int32_t* skipped,
                                              int32_t lag_index, int32_t count_skip, int32_t zero_grads, float* scale, int32_t* growth_tracker,
                                              uint32_t* done_counter, double growth_factor, double backoff_factor, int32_t growth_interval,
                                              tn_stream_t stream) {
  TN_REQUIRE(found_inf && scale && growth_tracker && done_counter && growth_interval >= 1, "tn_adam_step_ranges_amp_update: bad scaler arguments");
  TN_REQUIRE(num_ranges > 0, "tn_adam_step_ranges_amp_update: the fused scale update needs at least one range (use tn_grad_scaler_update otherwise)");
  ScalerUpdate su{scale, growth_tracker, done_counter, (float)growth_factor, (float)backoff_factor, (int32_t)growth_interval};
  return adam_ranges_amp_impl(params, grads, exp_avg, exp_avg_sq, num_ranges, offsets, counts, steps, lrs, lr_finals, sched_max_steps, sched_step, beta1, beta2,
                              eps, inv_scale, found_inf, flag_index, num_flags, skipped, lag_index, count_skip, zero_grads, su, stream);
}

int tn_adam_step_ranges_amp_update_cw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                                      const int64_t* counts, const int32_t* steps, const double* lrs, const double* lr_finals,
                                      const int32_t* sched_max_steps, int32_t sched_step, double beta1, double beta2, double eps, const float* inv_scale,
                                      float* found_inf, const int32_t* flag_index, int32_t num_flags, int32_t* skipped, int32_t lag_index,
                                      int32_t count_skip, int32_t zero_grads, float* scale, int32_t* growth_tracker, uint32_t* done_counter,
                                      double growth_factor, double backoff_factor, int32_t growth_interval, const TnSampleRays* next, bool* next_taken,
                                      tn_stream_t stream, const TnNextSamplingHost* chain, bool su_update) {
  TN_REQUIRE(found_inf && scale && growth_tracker && done_counter && growth_interval >= 1, "tn_adam_step_ranges_amp_update: bad scaler arguments");
  TN_REQUIRE(num_ranges > 0, "tn_adam_step_ranges_amp_update: the fused scale update needs at least one range (use tn_grad_scaler_update otherwise)");
  ScalerUpdate su{scale, growth_tracker, done_counter, (float)growth_factor, (float)backoff_factor, (int32_t)growth_interval};
  return adam_ranges_amp_impl(params, grads, exp_avg, exp_avg_sq, num_ranges, offsets, counts, steps, lrs, lr_finals, sched_max_steps, sched_step, beta1, beta2,
                              eps, inv_scale, found_inf, flag_index, num_flags, skipped, lag_index, count_skip, zero_grads, su_update ? su : ScalerUpdate{}, stream,
                              next, next_taken, chain);
}

// GradScaler.update() (torch/amp/grad_scaler.py -> amp_update_scale_cuda_kernel) for the fused step, one thread: backoff when any of the
// num_flags found_inf entries is set (and the schedule lag grows by one: the trainer does not step the LR schedulers then,
// engine/trainer.py:491-495), growth after `growth_interval` clean iterations in a row.
__global__ void k_grad_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, int num_flags, int32_t* lag, float growth_factor,
                                     float backoff_factor, int growth_interval, int clear) {
  scaler_update_body(scale, growth_tracker, found_inf, num_flags, lag, growth_factor, backoff_factor, growth_interval, clear);
}
extern "C" int tn_grad_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, int32_t num_flags, int32_t* lag,
                                     double growth_factor, double backoff_factor, int32_t growth_interval, int32_t clear_found_inf, tn_stream_t stream) {
  TN_REQUIRE(scale && growth_tracker && found_inf && num_flags >= 1 && growth_interval >= 1, "tn_grad_scaler_update: bad argument");
  hipLaunchKernelGGL(k_grad_scaler_update, dim3(1), dim3(1), 0, tn_s(stream), scale, growth_tracker, found_inf, (int)num_flags, lag, (float)growth_factor,
                     (float)backoff_factor, (int)growth_interval, (int)clear_found_inf);
  TN_CHECK_LAUNCH("tn_grad_scaler_update");
  return TN_OK;
}
