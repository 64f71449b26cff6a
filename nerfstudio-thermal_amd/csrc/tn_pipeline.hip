// tn_render_rays_eval: the no-grad render of one branch -- ThermalNerfactoModel.get_outputs in eval mode for one sampler + field
// (models/nerfacto.py:299-353: proposal_sampler -> field -> get_weights -> renderers; model_components/ray_samplers.py:577-618) -- as ONE
// call of the C ABI: the seven launches below are enqueued back to back by the library, nothing returns to the caller in between.
// Every stage is the entry point a caller could also invoke on its own (same kernels, same results, bit for bit).
#include "tn_common.h"

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
                                   const float* lin_pdf2, void* workspace, float* rgb, float* accumulation, float* depth_median,
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
  if ((rc = tn_field_pack_weights(field, w.field_ws, stream))) return rc;
  if ((rc = tn_field_fwd(field, origins, directions, camera_indices, e2, N, S2, 0, w.field_ws, density, rgb_s, nullptr, stream))) return rc;
  return tn_render_fwd(e2, density, rgb_s, N, S2, C, 0, w.w2, rgb, accumulation, depth_median, depth_expected, w.scratch, stream);
}
