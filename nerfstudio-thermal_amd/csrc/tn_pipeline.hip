// tn_render_rays_eval: the no-grad render of one branch -- ThermalNerfactoModel.get_outputs in eval mode for one sampler + field
// (models/nerfacto.py:299-353: proposal_sampler -> field -> get_weights -> renderers; model_components/ray_samplers.py:577-618) -- as ONE
// call of the C ABI: the seven launches below are enqueued back to back by the library, nothing returns to the caller in between.
// Every stage is the entry point a caller could also invoke on its own (same kernels, same results, bit for bit).
#include "tn_common.h"
#include "tn_pose_finish.h"

namespace {
struct EvalWs {
  float *s0, *e0, *d0, *w0, *s1, *e1, *d1, *w1, *s2, *e2, *w2, *rgb_s, *scratch;
  void* field_ws;
  int64_t bytes;
};
EvalWs eval_layout(void* base, int64_t N, int S0, int S1, int S2, int C) {
  EvalWs w;
  char* p = reinterpret_cast<char*>(base);
  int64_t off = 0;
  auto take = [&](int64_t floats) {
    float* r = reinterpret_cast<float*>(p + off);
    off += ((floats * 4 + 255) / 256) * 256;
    return r;
  };
  w.s0 = take(N * (S0 + 1)); w.e0 = take(N * (S0 + 1)); w.d0 = take(N * S0); w.w0 = take(N * S0);
  w.s1 = take(N * (S1 + 1)); w.e1 = take(N * (S1 + 1)); w.d1 = take(N * S1); w.w1 = take(N * S1);
  w.s2 = take(N * (S2 + 1)); w.e2 = take(N * (S2 + 1)); w.w2 = take(N * S2);
  w.rgb_s = take(N * (int64_t)S2 * C);
  w.scratch = take(TN_RENDER_SCRATCH_FLOATS);
  w.field_ws = p + off;
  off += tn_field_workspace_bytes(N * (int64_t)S2, 0);
  w.bytes = off;
  return w;
}
}  // namespace

extern "C" int64_t tn_render_rays_eval_workspace_bytes(int64_t num_rays, int32_t S0, int32_t S1, int32_t S2, int32_t C) {
  if (num_rays < 0 || S0 < 1 || S1 < 1 || S2 < 1 || S0 > TN_MAX_SAMPLES || S1 > TN_MAX_SAMPLES || S2 > TN_MAX_SAMPLES || C < 1 || C > 4) return TN_EINVAL;
  return eval_layout(nullptr, num_rays, S0, S1, S2, C).bytes;
}

extern "C" int tn_render_rays_eval(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* origins,
                                   const float* directions, const int64_t* camera_indices, const float* nears, const float* fars, int64_t N,
                                   int32_t S0, int32_t S1, int32_t S2, float anneal, const float* lin_spaced0, const float* lin_pdf1,
                                   const float* lin_pdf2, void* workspace, int64_t workspace_bytes, float* rgb, float* accumulation, float* depth_median,
                                   float* depth_expected, float* prop_depth0, float* prop_depth1, float* density, float* e_bins_out,
                                   float* rgb_samples_out, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(prop0 && prop1 && field && origins && directions && camera_indices && nears && fars && lin_spaced0 && lin_pdf1 && lin_pdf2 &&
                 workspace && rgb && density,
             "tn_render_rays_eval: null pointer");
  TN_REQUIRE(N > 0 && S0 >= 1 && S1 >= 1 && S2 >= 1 && S0 <= TN_MAX_SAMPLES && S1 <= TN_MAX_SAMPLES && S2 <= TN_MAX_SAMPLES,
             "tn_render_rays_eval: bad N=%lld S=(%d, %d, %d)", (long long)N, S0, S1, S2);
  TN_REQUIRE(((uintptr_t)workspace % 256) == 0, "tn_render_rays_eval: workspace must be 256-byte aligned");
  const int C = field->num_channels;
  EvalWs w = eval_layout(workspace, N, S0, S1, S2, C);
  TN_REQUIRE(workspace_bytes >= w.bytes, "tn_render_rays_eval: workspace of %lld bytes, tn_render_rays_eval_workspace_bytes = %lld", (long long)workspace_bytes,
             (long long)w.bytes);
  const int64_t field_ws_bytes = tn_field_workspace_bytes(N * (int64_t)S2, 0);
  float* e2 = e_bins_out ? e_bins_out : w.e2;
  float* rgb_s = rgb_samples_out ? rgb_samples_out : w.rgb_s;
  int rc;
  // ProposalNetworkSampler.generate_ray_samples: uniform bins -> density_fn -> get_weights + PDF resample, twice (no jitter at inference)
  if ((rc = tn_spaced_bins(lin_spaced0, nullptr, nears, fars, N, S0, w.s0, w.e0, stream))) return rc;
  if ((rc = tn_prop_density_fwd(prop0, origins, directions, w.e0, N, S0, w.d0, stream))) return rc;
  if ((rc = tn_weights_resample(w.e0, w.d0, w.s0, S0, anneal, lin_pdf1, nullptr, nears, fars, N, S1, w.w0, prop_depth0, w.s1, w.e1, stream))) return rc;
  if ((rc = tn_prop_density_fwd(prop1, origins, directions, w.e1, N, S1, w.d1, stream))) return rc;
  if ((rc = tn_weights_resample(w.e1, w.d1, w.s1, S1, anneal, lin_pdf2, nullptr, nears, fars, N, S2, w.w1, prop_depth1, w.s2, e2, stream))) return rc;
  // field (mean appearance embedding at inference), then get_weights + the renderers
  // (weight packing inside the field's first launch)
  if ((rc = tn_field_fwd_ex(field, origins, directions, camera_indices, e2, N, S2, 0, w.field_ws, field_ws_bytes, density, rgb_s, nullptr, 1, nullptr, 0, stream)))
    return rc;
  return tn_render_fwd(e2, density, rgb_s, N, S2, C, 0, w.w2, rgb, accumulation, depth_median, depth_expected, w.scratch, stream);
}

// ---- the TRAINING forward of one branch as one call: pose correction + level-0 bins, the two proposal levels (density -> weights + PDF
// resampling with jitter), the field with its activations kept for tn_field_bwd, get_weights + renderers.  Everything a backward pass or a
// loss needs later is written into ONE caller-provided buffer; tn_render_rays_train_layout gives the float offset of every tensor in it.
// Slots of the offsets array:
enum {
  TRO_ORIGINS = 0, TRO_DIRECTIONS,                       // [N,3] pose-corrected rays (the inputs themselves when pose_adjustment == NULL)
  TRO_S0, TRO_E0, TRO_D0, TRO_W0, TRO_M0,                // level 0: s_bins, e_bins [N,S0+1], density, weights [N,S0], median depth [N]
  TRO_S1, TRO_E1, TRO_D1, TRO_W1, TRO_M1,                // level 1
  TRO_S2, TRO_E2, TRO_D2, TRO_W2,                        // level 2 (field)
  TRO_RGB_SAMPLES, TRO_COMP, TRO_ACC, TRO_DEPTH, TRO_EXPECTED, TRO_SCRATCH,
  TRO_PENC0, TRO_PENC1,                                  // [N*S0][10], [N*S1][10]: the proposal levels' encodings (written with save_prop_enc only)
  TRO_END,
  TRO_COUNT
};
static void train_layout(int64_t N, int S0, int S1, int S2, int C, int64_t* off) {
  int64_t o = 0;
  auto take = [&](int slot, int64_t floats) { off[slot] = o; o += (floats + 63) / 64 * 64; };  // 256-byte aligned regions
  take(TRO_ORIGINS, N * 3); take(TRO_DIRECTIONS, N * 3);
  take(TRO_S0, N * (S0 + 1)); take(TRO_E0, N * (S0 + 1)); take(TRO_D0, N * S0); take(TRO_W0, N * S0); take(TRO_M0, N);
  take(TRO_S1, N * (S1 + 1)); take(TRO_E1, N * (S1 + 1)); take(TRO_D1, N * S1); take(TRO_W1, N * S1); take(TRO_M1, N);
  take(TRO_S2, N * (S2 + 1)); take(TRO_E2, N * (S2 + 1)); take(TRO_D2, N * S2); take(TRO_W2, N * S2);
  take(TRO_RGB_SAMPLES, N * (int64_t)S2 * C); take(TRO_COMP, N * C); take(TRO_ACC, N); take(TRO_DEPTH, N); take(TRO_EXPECTED, N);
  take(TRO_SCRATCH, TN_RENDER_SCRATCH_FLOATS);
  take(TRO_PENC0, N * (int64_t)S0 * 10); take(TRO_PENC1, N * (int64_t)S1 * 10);
  off[TRO_END] = o;
}

extern "C" int tn_render_rays_train_layout(int64_t num_rays, int32_t S0, int32_t S1, int32_t S2, int32_t C, int64_t* offsets, int32_t num_offsets) {
  TN_REQUIRE(offsets != nullptr && num_offsets >= TRO_COUNT, "tn_render_rays_train_layout: offsets needs %d entries", (int)TRO_COUNT);
  TN_REQUIRE(num_rays >= 0 && S0 >= 1 && S1 >= 1 && S2 >= 1 && S0 <= TN_MAX_SAMPLES && S1 <= TN_MAX_SAMPLES && S2 <= TN_MAX_SAMPLES && C >= 1 && C <= 4,
             "tn_render_rays_train_layout: bad argument");
  train_layout(num_rays, S0, S1, S2, C, offsets);
  return TN_OK;
}

// with_render = false: everything up to and including the field's forward (tn_train_step renders inside tn_render_losses_bwd)
static int render_rays_train_impl(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* pose_adjustment,
                                  const uint8_t* frozen, int32_t num_cameras, const float* origins_in, const float* directions_in,
                                  const int64_t* camera_indices, const float* nears, const float* fars, int64_t N, int32_t S0, int32_t S1,
                                  int32_t S2, float anneal, const float* jitter0, const float* jitter1, const float* jitter2,
                                  const float* lin_spaced0, const float* lin_pdf1, const float* lin_pdf2, void* field_workspace,
                                  int64_t field_workspace_bytes, float* out, void* wait_event_before_field, void* zero_fill, int64_t zero_fill_bytes,
                                  int32_t save_prop_enc, int with_render, int* clip_nblk, tn_stream_t stream, bool sampling_done = false) {
  if (clip_nblk) *clip_nblk = 0;
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(prop0 && prop1 && field && origins_in && directions_in && camera_indices && nears && fars && lin_spaced0 && lin_pdf1 && lin_pdf2 &&
                 field_workspace && out,
             "tn_render_rays_train: null pointer");
  TN_REQUIRE(N > 0 && S0 >= 1 && S1 >= 1 && S2 >= 1 && S0 <= TN_MAX_SAMPLES && S1 <= TN_MAX_SAMPLES && S2 <= TN_MAX_SAMPLES,
             "tn_render_rays_train: bad N=%lld S=(%d, %d, %d)", (long long)N, S0, S1, S2);
  TN_REQUIRE(((uintptr_t)out % 256) == 0, "tn_render_rays_train: the output buffer must be 256-byte aligned");
  const int C = field->num_channels;
  int64_t off[TRO_COUNT];
  train_layout(N, S0, S1, S2, C, off);
  auto at = [&](int slot) { return out + off[slot]; };
  const float* o = origins_in;
  const float* d = directions_in;
  int rc;
  if (sampling_done) {
    // the previous iteration's optimiser launch ran this batch's sampling front in its co-work blocks (TnTrainStep::next_sampling): pose-corrected
    // rays, all three levels' bins, both proposal levels' densities / weights / median depths (and encodings) are in `out` already
    if (pose_adjustment != nullptr) { o = at(TRO_ORIGINS); d = at(TRO_DIRECTIONS); }
  } else {
    if (pose_adjustment != nullptr) {  // CameraOptimizer.apply_to_raybundle and the first sampler level: independent, one launch
      TN_REQUIRE(num_cameras >= 1, "tn_render_rays_train: bad num_cameras=%d", num_cameras);
      if ((rc = tn_pose_spaced_bins(pose_adjustment, frozen, camera_indices, origins_in, directions_in, N, num_cameras, at(TRO_ORIGINS), at(TRO_DIRECTIONS),
                                    lin_spaced0, jitter0, nears, fars, S0, at(TRO_S0), at(TRO_E0), stream)))
        return rc;
      o = at(TRO_ORIGINS);
      d = at(TRO_DIRECTIONS);
    } else if ((rc = tn_spaced_bins(lin_spaced0, jitter0, nears, fars, N, S0, at(TRO_S0), at(TRO_E0), stream))) {
      return rc;
    }
    // save_prop_enc: the proposal networks take a gradient this iteration -- their encodings are kept for tn_render_rays_train_bwd
    if ((rc = tn_prop_density_fwd_ex(prop0, o, d, at(TRO_E0), N, S0, at(TRO_D0), save_prop_enc ? at(TRO_PENC0) : nullptr, stream))) return rc;
    if ((rc = tn_weights_resample(at(TRO_E0), at(TRO_D0), at(TRO_S0), S0, anneal, lin_pdf1, jitter1, nears, fars, N, S1, at(TRO_W0), at(TRO_M0), at(TRO_S1),
                                  at(TRO_E1), stream)))
      return rc;
    if ((rc = tn_prop_density_fwd_ex(prop1, o, d, at(TRO_E1), N, S1, at(TRO_D1), save_prop_enc ? at(TRO_PENC1) : nullptr, stream))) return rc;
    if ((rc = tn_weights_resample(at(TRO_E1), at(TRO_D1), at(TRO_S1), S1, anneal, lin_pdf2, jitter2, nears, fars, N, S2, at(TRO_W1), at(TRO_M1), at(TRO_S2),
                                  at(TRO_E2), stream)))
      return rc;
  }
  if (wait_event_before_field != nullptr) {
    // the previous iteration's Adam launch over the field's parameters may still be running on another stream (it overlaps the proposal
    // sampling above, which only reads the proposal networks): the field's first read of its parameters waits for it here
    hipError_t e = hipStreamWaitEvent(tn_s(stream), reinterpret_cast<hipEvent_t>(wait_event_before_field), 0);
    TN_REQUIRE(e == hipSuccess, "tn_render_rays_train: hipStreamWaitEvent failed: %s", hipGetErrorString(e));
  }
  // weight packing and the zero-fill of the caller's step accumulators ride in the field's first launch
  if ((rc = tn_field_fwd_ex(field, o, d, camera_indices, at(TRO_E2), N, S2, 1, field_workspace, field_workspace_bytes, at(TRO_D2), at(TRO_RGB_SAMPLES), nullptr, 1,
                            zero_fill, zero_fill_bytes, stream)))
    return rc;
  if (!with_render) return TN_OK;  // (with_render: 0 no renderers, 1 tn_render_fwd, 2 tn_render_fwd without its clip launch)
  return tn_render_fwd_ex(at(TRO_E2), at(TRO_D2), at(TRO_RGB_SAMPLES), N, S2, C, 1, at(TRO_W2), at(TRO_COMP), at(TRO_ACC), at(TRO_DEPTH), at(TRO_EXPECTED),
                          at(TRO_SCRATCH), with_render == 1, clip_nblk, stream);
}
extern "C" int tn_render_rays_train(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* pose_adjustment,
                                    const uint8_t* frozen, int32_t num_cameras, const float* origins_in, const float* directions_in,
                                    const int64_t* camera_indices, const float* nears, const float* fars, int64_t N, int32_t S0, int32_t S1,
                                    int32_t S2, float anneal, const float* jitter0, const float* jitter1, const float* jitter2,
                                    const float* lin_spaced0, const float* lin_pdf1, const float* lin_pdf2, void* field_workspace,
                                    int64_t field_workspace_bytes, float* out, void* wait_event_before_field, void* zero_fill, int64_t zero_fill_bytes,
                                    int32_t save_prop_enc, tn_stream_t stream) {
  return render_rays_train_impl(prop0, prop1, field, pose_adjustment, frozen, num_cameras, origins_in, directions_in, camera_indices, nears, fars, N, S0, S1, S2,
                                anneal, jitter0, jitter1, jitter2, lin_spaced0, lin_pdf1, lin_pdf2, field_workspace, field_workspace_bytes, out,
                                wait_event_before_field, zero_fill, zero_fill_bytes, save_prop_enc, 1, nullptr, stream);
}

// ---- the TRAINING backward of one branch as one call: everything behind d(composite) / d(weights) of the losses -- tn_render_bwd (get_weights +
// renderers), the field's backward (MLP chain + weight gradients, d position, table scatter) and, when the proposal networks take a gradient
// this iteration, tn_weights_bwd + tn_prop_density_bwd of both levels on two companion streams of `stream` (forked after the renderer backward,
// joined before returning) -- enqueued by the library: the same launches, in the same order per stream, as the caller would make one by one
// (nerfstudio_thermal_amd/engine.py: loss_and_backward), without ~15 host round trips through the binding.
__global__ void k_add_inplace(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] += b[i];
}
extern "C" int64_t tn_render_rays_train_bwd_tmp_floats(int64_t N, int32_t S0, int32_t S1, int32_t S2, int32_t C) {
  if (N < 0 || S0 < 1 || S1 < 1 || S2 < 1 || C < 1 || C > 4) return TN_EINVAL;
  auto up = [](int64_t x) { return (x + 63) / 64 * 64; };
  return up(N * (int64_t)S2 * C) + up(N * (int64_t)S2) + up(N * (int64_t)S0) + up(N * (int64_t)S1);
}
// render_bwd_done: d_rgb / d_density at the head of `tmp` are there already (tn_render_losses_bwd wrote them)
static int render_rays_train_bwd_impl(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* origins,
                                      const float* directions, const int64_t* camera_indices, int64_t N, int32_t S0, int32_t S1, int32_t S2,
                                      const float* fwd_out, const float* d_comp, const float* d_weights0,
                                      const float* d_weights1, const float* d_weights2, const float* d_density_extra, void* field_workspace,
                                      int64_t field_workspace_bytes, void* prop_workspace0, int64_t prop_workspace_bytes0, void* prop_workspace1,
                                      int64_t prop_workspace_bytes1, float* tmp, float* d_origins, float* d_directions, int32_t prop_enc_saved,
                                      bool render_bwd_done, const PoseFinishArgs* fold_cowork, bool* fold_cowork_taken, tn_stream_t stream) {
  if (fold_cowork_taken) *fold_cowork_taken = false;
  if (N == 0) return TN_OK;
  TN_REQUIRE(field && origins && directions && camera_indices && fwd_out && d_comp && d_weights2 && field_workspace && tmp,
             "tn_render_rays_train_bwd: null pointer");
  TN_REQUIRE((d_weights0 == nullptr) == (d_weights1 == nullptr), "tn_render_rays_train_bwd: d_weights0 and d_weights1 come together (or not at all)");
  const bool prop_grad = d_weights0 != nullptr;
  if (prop_grad) TN_REQUIRE(prop0 && prop1 && prop_workspace0 && prop_workspace1, "tn_render_rays_train_bwd: proposal networks / workspaces missing");
  TN_REQUIRE(N > 0 && S0 >= 1 && S1 >= 1 && S2 >= 1 && S0 <= TN_MAX_SAMPLES && S1 <= TN_MAX_SAMPLES && S2 <= TN_MAX_SAMPLES,
             "tn_render_rays_train_bwd: bad N=%lld S=(%d, %d, %d)", (long long)N, S0, S1, S2);
  TN_REQUIRE((d_origins == nullptr) == (d_directions == nullptr), "tn_render_rays_train_bwd: d_origins and d_directions must both be given or both NULL");
  const int C = field->num_channels;
  int64_t off[TRO_COUNT];
  train_layout(N, S0, S1, S2, C, off);
  auto at = [&](int slot) { return fwd_out + off[slot]; };
  auto up = [](int64_t x) { return (x + 63) / 64 * 64; };
  float* d_rgb = tmp;
  float* d_dens = d_rgb + up(N * (int64_t)S2 * C);
  float* dd0 = d_dens + up(N * (int64_t)S2);
  float* dd1 = dd0 + up(N * (int64_t)S0);
  const float* o = origins;  // the pose-corrected rays the forward used (its TRO_ORIGINS / TRO_DIRECTIONS slots, or its inputs without a pose)
  const float* d = directions;
  hipStream_t st = tn_s(stream);
  int rc;
  if (!render_bwd_done && (rc = tn_render_bwd(at(TRO_E2), at(TRO_D2), at(TRO_RGB_SAMPLES), at(TRO_W2), d_comp, d_weights2, N, S2, C, d_rgb, d_dens, stream)))
    return rc;
  if (d_density_extra != nullptr) {  // the density loss's gradient on this branch's own density (separate mode)
    const int64_t n = N * (int64_t)S2;
    hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)std::min<int64_t>(tn_cdiv(n, 256), 4096)), dim3(256), 0, st, d_dens, d_density_extra, n);
    TN_CHECK_LAUNCH("tn_render_rays_train_bwd(add)");
  }
  int rc0 = TN_OK, rc1 = TN_OK;
  if (prop_grad) {
    // the proposal networks' backward (own tables, MLPs, scatter; d origins / d directions accumulated atomically) is independent of the field's
    hipStream_t s0 = tn_fork_n(st, 1), s1 = tn_fork_n(st, 2);
    tn_stream_t t0 = s0 ? (tn_stream_t)s0 : stream, t1 = s1 ? (tn_stream_t)s1 : stream;
    rc0 = tn_weights_bwd(at(TRO_E0), at(TRO_D0), at(TRO_W0), d_weights0, N, S0, dd0, t0);
    if (!rc0)
      rc0 = tn_prop_density_bwd_ex(prop0, o, d, at(TRO_E0), dd0, N, S0, prop_workspace0, prop_workspace_bytes0, d_origins, d_directions,
                                   prop_enc_saved ? at(TRO_PENC0) : nullptr, t0);
    rc1 = tn_weights_bwd(at(TRO_E1), at(TRO_D1), at(TRO_W1), d_weights1, N, S1, dd1, t1);
    if (!rc1)
      rc1 = tn_prop_density_bwd_ex(prop1, o, d, at(TRO_E1), dd1, N, S1, prop_workspace1, prop_workspace_bytes1, d_origins, d_directions,
                                   prop_enc_saved ? at(TRO_PENC1) : nullptr, t1);
  }
  // (d position forks to its companion stream only when the proposal networks' backward keeps other queues busy anyway)
  // (the fold's co-work only on iterations without the proposal networks' backward: their contributions to d origins / d directions arrive with the join below)
  rc = tn_field_bwd_phase_ex(field, o, d, camera_indices, at(TRO_E2), d_dens, d_rgb, N, S2, field_workspace, field_workspace_bytes, d_origins, d_directions,
                             TN_BWD_MLP | TN_BWD_SCATTER | TN_BWD_JOIN | (prop_grad ? TN_BWD_FORK_DPOS : 0), 0, field->grid.num_levels,
                             prop_grad ? nullptr : fold_cowork, fold_cowork_taken, stream);
  if (prop_grad) { tn_join_n(st, 1); tn_join_n(st, 2); }
  return rc ? rc : (rc0 ? rc0 : rc1);
}
extern "C" int tn_render_rays_train_bwd(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* origins,
                                        const float* directions, const int64_t* camera_indices, int64_t N, int32_t S0, int32_t S1, int32_t S2,
                                        const float* fwd_out, const float* d_comp, const float* d_weights0,
                                        const float* d_weights1, const float* d_weights2, const float* d_density_extra, void* field_workspace,
                                        int64_t field_workspace_bytes, void* prop_workspace0, int64_t prop_workspace_bytes0, void* prop_workspace1,
                                        int64_t prop_workspace_bytes1, float* tmp, float* d_origins, float* d_directions, int32_t prop_enc_saved,
                                        tn_stream_t stream) {
  return render_rays_train_bwd_impl(prop0, prop1, field, origins, directions, camera_indices, N, S0, S1, S2, fwd_out, d_comp, d_weights0, d_weights1, d_weights2,
                                    d_density_extra, field_workspace, field_workspace_bytes, prop_workspace0, prop_workspace_bytes0, prop_workspace1,
                                    prop_workspace_bytes1, tmp, d_origins, d_directions, prop_enc_saved, false, nullptr, nullptr, stream);
}


// tn_train_step: one training iteration of the shared-density model as ONE call (see the header): the five entry points RenderEngine.train_step
// calls, in its order, on the caller's stream.  Nothing here launches a kernel of its own.
extern "C" int tn_train_step(const TnTrainStep* a, tn_stream_t stream) {
  TN_REQUIRE(a != nullptr, "tn_train_step: null argument block");
  tn_join_n(tn_s(stream), 3);  // (TN_NEXT_SAMPLING=3: the previous call's sampling front ran on a companion stream; a no-op otherwise)
  if (a->next_sample_taken) *a->next_sample_taken = 0;  // (set once the optimiser launch carries it: any earlier return leaves the batch to the caller)
  if (a->N == 0) return TN_OK;
  TN_REQUIRE(a->prop0 && a->prop1 && a->field && a->origins_in && a->directions_in && a->camera_indices && a->image && a->is_thermal && a->nears &&
                 a->fars && a->fwd_out && a->bwd_tmp && a->acc && a->losses16 && a->loss_lines && a->d_comp && a->d_weights2,
             "tn_train_step: null pointer");
  TN_REQUIRE(a->pose_adjustment && a->grad_pose && a->d_origins && a->d_directions && a->num_cameras >= 1,
             "tn_train_step: this entry point is the iteration WITH a camera optimiser (pose_adjustment, grad_pose, d_origins, d_directions)");
  TN_REQUIRE(a->prop_grad == 0 || (a->d_weights0 && a->d_weights1 && a->prop_workspace0 && a->prop_workspace1),
             "tn_train_step: prop_grad needs d_weights0 / d_weights1 and both proposal workspaces");
  TN_REQUIRE(a->num_check >= 0 && a->num_check <= TN_TRAIN_STEP_MAX_RANGES && a->num_ranges >= 0 && a->num_ranges <= TN_TRAIN_STEP_MAX_RANGES,
             "tn_train_step: at most %d ranges", TN_TRAIN_STEP_MAX_RANGES);
  TN_REQUIRE(a->found_inf && a->num_flags >= 1 && a->scale && a->growth_tracker && a->done_counter && a->skipped,
             "tn_train_step: the GradScaler state is part of the iteration (found_inf, scale, growth_tracker, done_counter, skipped)");
  TN_REQUIRE(a->num_ranges == 0 || (a->params && a->grads && a->exp_avg && a->exp_avg_sq), "tn_train_step: null arena pointer");
  const int C = a->field->num_channels;
  TN_REQUIRE(C == 4, "tn_train_step: the shared-density model renders RGB + thermal from one field (4 channels), got %d", C);
  // Everything a LATER stage would refuse is refused HERE, before the first launch: sizes of the three workspaces, sample counts, alignment of the
  // forward's buffer and of the accumulator block -- so an error return means nothing was enqueued (the caller's step counters then still match the
  // device's; the inner entry points keep their own checks).
  TN_REQUIRE(a->N > 0 && a->S0 >= 1 && a->S1 >= 1 && a->S2 >= 1 && a->S0 <= TN_MAX_SAMPLES && a->S1 <= TN_MAX_SAMPLES && a->S2 <= TN_MAX_SAMPLES,
             "tn_train_step: bad N=%lld S=(%d, %d, %d)", (long long)a->N, a->S0, a->S1, a->S2);
  TN_REQUIRE(((uintptr_t)a->fwd_out % 256) == 0, "tn_train_step: the forward's buffer must be 256-byte aligned");
  TN_REQUIRE(((uintptr_t)a->acc % 16) == 0 && a->acc_bytes >= 0 && a->acc_bytes % 16 == 0, "tn_train_step: the accumulator block must be 16-byte aligned and sized");
  TN_REQUIRE(a->lin_spaced0 && a->lin_pdf1 && a->lin_pdf2 && a->field_workspace, "tn_train_step: null sampler table / field workspace");
  {
    const int64_t need_f = tn_field_workspace_bytes(a->N * (int64_t)a->S2, 1);
    TN_REQUIRE(a->field_workspace_bytes >= need_f, "tn_train_step: field workspace of %lld bytes, tn_field_workspace_bytes(%lld, 1) = %lld",
               (long long)a->field_workspace_bytes, (long long)(a->N * (int64_t)a->S2), (long long)need_f);
    if (a->prop_grad) {
      const int64_t need0 = tn_prop_workspace_bytes(a->N * (int64_t)a->S0), need1 = tn_prop_workspace_bytes(a->N * (int64_t)a->S1);
      TN_REQUIRE(a->prop_workspace_bytes0 >= need0 && a->prop_workspace_bytes1 >= need1,
                 "tn_train_step: proposal workspaces of %lld / %lld bytes, tn_prop_workspace_bytes = %lld / %lld", (long long)a->prop_workspace_bytes0,
                 (long long)a->prop_workspace_bytes1, (long long)need0, (long long)need1);
    }
  }
  if (a->next_sampling_taken) *a->next_sampling_taken = 0;
  if (a->next_sample != nullptr && a->num_ranges > 0) {  // what the optimiser launch's co-work would refuse, refused here too
    const TnSampleRays* s = a->next_sample;
    TN_REQUIRE(s->num_rays >= 0 && s->num_images >= 1 && s->patch_size >= 1 && s->patch_size <= 8, "tn_train_step(next_sample): bad num_rays=%lld num_images=%d patch_size=%d",
               (long long)s->num_rays, s->num_images, s->patch_size);
    if (s->num_rays > 0) {
      const int64_t pp = (int64_t)s->patch_size * s->patch_size, per = ((s->num_rays / s->num_images) / pp) * pp, last = s->num_rays - (int64_t)(s->num_images - 1) * per;
      TN_REQUIRE(last > 0 && last % pp == 0, "tn_train_step(next_sample): %lld rays over %d images do not split into whole patches", (long long)s->num_rays, s->num_images);
      TN_REQUIRE(s->images && s->image_offsets && s->heights && s->widths && s->is_thermal && s->image_idx && s->u && s->ray_indices && s->image &&
                     s->is_thermal_out && s->c2w && s->fx && s->fy && s->cx && s->cy && s->origins && s->directions && s->num_cameras >= 1,
                 "tn_train_step(next_sample): null pointer / bad camera arguments");
    }
  }
  if (a->next_sampling != nullptr) TN_REQUIRE(a->next_sampling->fwd_out != nullptr && ((uintptr_t)a->next_sampling->fwd_out % 256) == 0,
                                              "tn_train_step(next_sampling): the next forward buffer must be given and 256-byte aligned");
  int64_t off[TRO_COUNT];
  train_layout(a->N, a->S0, a->S1, a->S2, C, off);
  int rc;
  // TN_FUSE_RENDER=1 (read per call): the renderers, the losses and the renderers' backward as ONE launch, tn_render_losses_bwd, instead of their
  // three entry points.  A measured experiment, off by default: these per-ray stages are bound by vector-instruction issue (4096 rays = 4 waves
  // per SIMD, ~4 k instructions per ray), not by launches or latency, so the one launch takes what the three take together (41 vs 37 us).
  const char* fr = getenv("TN_FUSE_RENDER");
  const bool fuse_render = fr && fr[0] == '1' && a->N % 4 == 0;
  int clip_nblk = 0;  // (the batch-wide clip of the expected depth rides in the loss launch: tn_train_losses_clip)
  if ((rc = render_rays_train_impl(a->prop0, a->prop1, a->field, a->pose_adjustment, a->frozen, a->num_cameras, a->origins_in, a->directions_in,
                                   a->camera_indices, a->nears, a->fars, a->N, a->S0, a->S1, a->S2, a->anneal, a->jitter0, a->jitter1, a->jitter2,
                                   a->lin_spaced0, a->lin_pdf1, a->lin_pdf2, a->field_workspace, a->field_workspace_bytes, a->fwd_out, nullptr, a->acc,
                                   a->acc_bytes, a->prop_grad ? 1 : 0, fuse_render ? 0 : 2, &clip_nblk, stream, a->sampling_done != 0)))
    return rc;
  float* out = a->fwd_out;
  const float* sprop[2] = {out + off[TRO_S0], out + off[TRO_S1]};
  const float* wprop[2] = {out + off[TRO_W0], out + off[TRO_W1]};
  const int32_t Sprop[2] = {a->S0, a->S1};
  float* dwprop[2] = {a->prop_grad ? a->d_weights0 : nullptr, a->prop_grad ? a->d_weights1 : nullptr};
  const float* comp = out + off[TRO_COMP];
  if (fuse_render) {
    float* d_rgb = a->bwd_tmp;  // the layout tn_render_rays_train_bwd gives its `tmp`
    float* d_dens = d_rgb + (a->N * (int64_t)a->S2 * C + 63) / 64 * 64;
    if ((rc = tn_render_losses_bwd(out + off[TRO_E2], out + off[TRO_D2], out + off[TRO_RGB_SAMPLES], a->N, a->S2, C, out + off[TRO_W2], out + off[TRO_COMP],
                                   out + off[TRO_ACC], out + off[TRO_DEPTH], out + off[TRO_EXPECTED], out + off[TRO_SCRATCH], out + off[TRO_S2], 2, sprop, wprop,
                                   Sprop, dwprop, a->distortion_mult, a->interlevel_mult, a->d_weights2, a->image, a->is_thermal, a->thermal_mult, a->tv_mult,
                                   a->cross_mult, a->d_comp, a->loss_lines, d_rgb, d_dens, 1, stream)))
      return rc;
  } else if ((rc = tn_train_losses_clip(out + off[TRO_S2], out + off[TRO_W2], a->S2, 2, sprop, wprop, Sprop, dwprop, a->N, a->distortion_mult,
                                        a->interlevel_mult, a->d_weights2, comp, C, comp + 3, C, a->image, a->is_thermal, a->thermal_mult, a->tv_mult,
                                        a->cross_mult, a->d_comp, a->d_comp + 3, a->loss_lines, out + off[TRO_EXPECTED], out + off[TRO_SCRATCH], clip_nblk,
                                        stream))) {
    // pixel terms on the RGB columns / the thermal column of the one RGBT composite (models/thermal_nerfacto.py:425-428)
    return rc;
  }
  // The launch that ends the backward (pose gradient, loss sums, regulariser, GradScaler's check of the small ranges) rides in the first blocks of
  // the main grid's fold launch on iterations without a proposal update (TN_POSE_FINISH_COWORK=0: always a launch of its own)
  PoseFinishArgs pf;
  bool pf_taken = false;
  const char* pfe = getenv("TN_POSE_FINISH_COWORK");
  const bool pf_cowork = !a->prop_grad && !(pfe && pfe[0] == '0');
  if (pf_cowork &&
      (rc = tn_pose_finish_args(a->pose_adjustment, a->frozen, a->camera_indices, a->directions_in, a->d_origins, a->d_directions, a->N, a->num_cameras,
                                a->grad_pose, a->loss_lines, a->losses16, a->trans_pen, a->rot_pen, a->pen_scale, a->losses16 + 11, a->grads, a->num_check,
                                a->check_offsets, a->check_counts, a->check_flags, a->num_flags, a->found_inf, a->pose_flag, pf)))
    return rc;
  if ((rc = render_rays_train_bwd_impl(a->prop_grad ? a->prop0 : nullptr, a->prop_grad ? a->prop1 : nullptr, a->field, out + off[TRO_ORIGINS],
                                       out + off[TRO_DIRECTIONS], a->camera_indices, a->N, a->S0, a->S1, a->S2, out, a->d_comp,
                                       a->prop_grad ? a->d_weights0 : nullptr, a->prop_grad ? a->d_weights1 : nullptr, a->d_weights2, nullptr,
                                       a->field_workspace, a->field_workspace_bytes, a->prop_grad ? a->prop_workspace0 : nullptr,
                                       a->prop_grad ? a->prop_workspace_bytes0 : 0, a->prop_grad ? a->prop_workspace1 : nullptr,
                                       a->prop_grad ? a->prop_workspace_bytes1 : 0, a->bwd_tmp, a->d_origins, a->d_directions, a->prop_grad ? 1 : 0,
                                       fuse_render, pf_cowork ? &pf : nullptr, &pf_taken, stream)))
    return rc;
  if (!pf_taken) {
    if ((rc = tn_pose_bwd_finish_check(a->pose_adjustment, a->frozen, a->camera_indices, a->directions_in, a->d_origins, a->d_directions, a->N,
                                       a->num_cameras, a->grad_pose, a->loss_lines, a->losses16, a->trans_pen, a->rot_pen, a->pen_scale, a->losses16 + 11,
                                       a->grads, a->num_check, a->check_offsets, a->check_counts, a->check_flags, a->num_flags, a->found_inf, a->pose_flag,
                                       stream)))
      return rc;
  }
  if (a->num_ranges == 0) return TN_OK;
  bool taken = false;
  // The next iteration's sampling front as co-work (TnTrainStep::next_sampling): the ranges it READS -- whatever holds the proposal networks or the
  // pose corrections -- are stepped first, in a launch of their own (a few MB); the launch that carries the chain steps the rest (the field).
  const TnNextSampling* nx = a->next_sampling;
  const char* nse = getenv("TN_NEXT_SAMPLING");
  bool chain = nx != nullptr && a->next_sample != nullptr && a->next_sample->num_rays == a->N && a->next_sample->camera_indices != nullptr && a->N % 4 == 0 &&
               tn_next_sampling_supported(a->S0, a->S1, a->S2) && !(nse && nse[0] == '0');
  int first[TN_TRAIN_STEP_MAX_RANGES], nfirst = 0, rest[TN_TRAIN_STEP_MAX_RANGES], nrest = 0;
  if (chain) {
    const float* reads[] = {a->prop0->grid.table, a->prop0->w0, a->prop0->b0, a->prop0->w1, a->prop0->b1, a->prop1->grid.table, a->prop1->w0,
                            a->prop1->b0, a->prop1->w1, a->prop1->b1, a->pose_adjustment};
    for (int k = 0; k < a->num_ranges; ++k) {
      const float *lo = a->params + a->offsets[k], *hi = lo + a->counts[k];
      bool hit = false;
      for (const float* q : reads) hit = hit || (q >= lo && q < hi);
      if (hit) first[nfirst++] = k; else if (a->counts[k] > 0) rest[nrest++] = k;
    }
    chain = nrest > 0 && nfirst > 0;  // (nothing left to run beside, or nothing to step first: one launch, no chain)
  }
  if (!chain) {
    rc = tn_adam_step_ranges_amp_update_cw(a->params, a->grads, a->exp_avg, a->exp_avg_sq, a->num_ranges, a->offsets, a->counts, a->steps, a->lrs, a->lr_finals,
                                           a->sched_max_steps, a->sched_step, a->beta1, a->beta2, a->eps, nullptr, a->found_inf, a->flag_index, a->num_flags,
                                           a->skipped, a->lag_index, 1, 1, a->scale, a->growth_tracker, a->done_counter, a->growth_factor, a->backoff_factor,
                                           a->growth_interval, a->next_sample, &taken, stream);
  } else {
    auto sub = [&](const int* idx, int n, int64_t* o, int64_t* c, int32_t* s, double* l, double* lf, int32_t* ms, int32_t* fl) {
      for (int i = 0; i < n; ++i) {
        const int k = idx[i];
        o[i] = a->offsets[k]; c[i] = a->counts[k]; s[i] = a->steps[k]; l[i] = a->lrs[k]; lf[i] = a->lr_finals[k]; ms[i] = a->sched_max_steps[k]; fl[i] = a->flag_index[k];
      }
    };
    int64_t o[TN_TRAIN_STEP_MAX_RANGES], c[TN_TRAIN_STEP_MAX_RANGES];
    int32_t s[TN_TRAIN_STEP_MAX_RANGES], ms[TN_TRAIN_STEP_MAX_RANGES], fl[TN_TRAIN_STEP_MAX_RANGES];
    double l[TN_TRAIN_STEP_MAX_RANGES], lf[TN_TRAIN_STEP_MAX_RANGES];
    rc = TN_OK;
    // (the first launch reads found_inf / the schedule lag like the second; GradScaler.update() comes with the second.  It also carries the batch's
    // pixel sampling + ray generation in its co-work row, as the one optimiser launch of other iterations does)
    sub(first, nfirst, o, c, s, l, lf, ms, fl);
    rc = tn_adam_step_ranges_amp_update_cw(a->params, a->grads, a->exp_avg, a->exp_avg_sq, nfirst, o, c, s, l, lf, ms, a->sched_step, a->beta1, a->beta2, a->eps,
                                           nullptr, a->found_inf, fl, a->num_flags, a->skipped, a->lag_index, 1, 1, a->scale, a->growth_tracker, a->done_counter,
                                           a->growth_factor, a->backoff_factor, a->growth_interval, a->next_sample, &taken, stream, nullptr, false);
    if (!rc && !taken) rc = tn_sample_rays_args(a->next_sample, stream);  // (no range of the first launch was live: the batch as a launch of its own)
    if (!rc) {
      taken = true;
      TnNextSamplingHost h{};
      h.prop0 = a->prop0; h.prop1 = a->prop1; h.pose = a->pose_adjustment; h.frozen = a->frozen; h.num_cameras = a->num_cameras;
      h.rays_o = a->next_sample->origins; h.rays_d = a->next_sample->directions; h.cam = a->next_sample->camera_indices;
      h.nears = a->nears; h.fars = a->fars; h.jit0 = nx->jitter0; h.jit1 = nx->jitter1; h.jit2 = nx->jitter2;
      h.lin0 = a->lin_spaced0; h.lin1 = a->lin_pdf1; h.lin2 = a->lin_pdf2; h.anneal = nx->anneal;
      h.S0 = a->S0; h.S1 = a->S1; h.S2 = a->S2; h.N = a->N;
      h.out = nx->fwd_out;
      const int slots[16] = {TRO_ORIGINS, TRO_DIRECTIONS, TRO_S0, TRO_E0, TRO_D0, TRO_W0, TRO_M0, TRO_S1, TRO_E1, TRO_D1, TRO_W1, TRO_M1, TRO_S2, TRO_E2, TRO_PENC0,
                             TRO_PENC1};
      for (int k = 0; k < 16; ++k) h.off[k] = off[slots[k]];
      h.save_enc = nx->prop_grad ? 1 : 0;
      sub(rest, nrest, o, c, s, l, lf, ms, fl);
      rc = tn_adam_step_ranges_amp_update_cw(a->params, a->grads, a->exp_avg, a->exp_avg_sq, nrest, o, c, s, l, lf, ms, a->sched_step, a->beta1, a->beta2, a->eps,
                                             nullptr, a->found_inf, fl, a->num_flags, a->skipped, a->lag_index, 1, 1, a->scale, a->growth_tracker, a->done_counter,
                                             a->growth_factor, a->backoff_factor, a->growth_interval, nullptr, nullptr, stream, &h, true);
      if (!rc && a->next_sampling_taken) *a->next_sampling_taken = 1;
    }
  }
  if (a->next_sample_taken) *a->next_sample_taken = taken ? 1 : 0;
  return rc;
}
