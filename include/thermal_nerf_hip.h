/*
 * thermal_nerf_hip.h -- C ABI of libthermal_nerf_hip.so (MI355X / gfx950).
 *
 * The reference (yvette256/nerfstudio-thermal) has no FFI of its own: its native arithmetic enters
 * through the Python objects of tinycudann (tcnn.Encoding / tcnn.Network / tcnn.NetworkWithInputEncoding)
 * and, on the CPU path that is this project's oracle, through ATen.  Each entry point below replaces one
 * such seam; the reference location it stands in for is cited as (file:line), relative to
 * /root/reference/nerfstudio/.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *  - plain C: device pointers + sizes + an opaque stream handle (a hipStream_t passed as void*; NULL = default stream).
 *  - every function returns 0 on success or a negative TN_E* code; nothing throws, nothing allocates or frees
 *    caller memory, nothing synchronises the device.  Work is enqueued on `stream`.
 *  - all float tensors are fp32, contiguous, row-major; sample tensors are dense [N, S] per level
 *    (N rays, S samples; "bins" tensors are [N, S+1]).
 *  - operand-shape violations are rejected on the host (TN_EINVAL) before anything is launched.
 *
 * State and environment (everything the library keeps or reads besides its arguments):
 *  - Process-wide state: ONE map of companion streams -- a second HIP stream (+ two events) per (device, caller stream), created the first time
 *    an entry point forks work beside the caller's stream (tn_field_bwd*: k_field_dpos beside the table scatter) and kept for the life of the
 *    process; tn_shutdown() waits for them and destroys them (they are re-created on demand).  The last error string (tn_last_error) is
 *    per process too.  There is no other hidden state: no context object, no caches, no allocations -- workspaces are the caller's.
 *    (SURVEY.md 8b proposed tn_create / tn_destroy around an opaque context; with the state reduced to this map a context would own nothing
 *    else, so the teardown is the one call.)
 *  - The data-parallel gradient exchange: the Python package uses torch.distributed over RCCL on slices of the caller's gradient arena
 *    (nerfstudio_thermal_amd/parallel.py), exactly where the reference has torch DDP; tn_comm_* / tn_allreduce_grads offer the same exchange to a
 *    host that binds only this ABI (RCCL resolved at run time, no link-time dependency).
 *  - Environment switches, read once per process, all tuning / diagnostic aids whose defaults are the product path:
 *      TN_NO_FORK=1               no companion streams (everything on the caller's stream)
 *      TN_SCATTER_MODE=2|1|0      table-gradient scatter: 2 = segmented (default: block-private record regions, no global atomics), 1 = binned
 *                                 (rounds 2-5: per-bucket arrays with reservations), 0 = the round-1 global-atomic scatter with dense replicas.
 *                                 Read on every call; the bin and fold launches of one backward must see the same value.
 *      TN_SCATTER_REPLICAS=n, TN_SCATTER_SPARSE_CHUNK=n, TN_SCATTER_MERGE_RES=n      tuning knobs of the two scatter paths
 *      TN_FOLD_TRACE=1 [TN_FOLD_TRACE_FILE=path]   per-block timing of the fold pass (synchronises and prints: diagnostics only)
 *      TN_BIN_LEVEL_GROUPS=n      force the number of level groups of the bin pass (diagnostic: n = levels -> one level per block)
 *      TN_DPOS_COWORK=0           (read per call) the main field's d position pass as a launch of its own (forked with TN_BWD_FORK_DPOS) instead of
 *                                 extra blocks of the table scatter's bin launch
 *      TN_POSE_FINISH_COWORK=0    (read per call) tn_train_step's last backward launch (tn_pose_bwd_finish_check) always as a launch of its own instead of
 *                                 the first blocks of the main grid's fold launch on iterations without a proposal update
 *      TN_FUSE_RENDER=1           (read per call) tn_train_step with tn_render_fwd / tn_train_losses / tn_render_bwd as ONE launch,
 *                                 tn_render_losses_bwd -- a measured experiment that is correct and not faster (profiles/r05_experiments.md)
 *      TN_NEXT_SAMPLING=0|2|3|4   (read per call) tn_train_step's next_sampling: 0 = never taken (every iteration samples in line); 2 = the chain as a launch
 *                                 of its own behind the optimiser launch; 3 = on a companion stream beside it (A/B timing aids: same results);
 *                                 4 = (opt-in experiment) the chain's waves also step most of the field's optimiser range themselves, between
 *                                 their stages, through LDS-staged asynchronous loads: same forward buffer bit for bit, the same Adam
 *                                 arithmetic; the launch 141 -> 132 us (profiles/r06_next_sampling.md).  TN_FUSED_SITES=<n> sizes its share
 *      TN_HEAD_BF16X3=1           (read per call; opt-in experiment, never the default) the colour head's two 64-wide layers in tn_field_fwd /
 *                                 tn_field_bwd* on split-bf16 matrix instructions: x = hi + lo in bf16, three v_mfma_f32_32x32x16_bf16 per
 *                                 product, fp32 accumulators (~2^-16 relative).  The density path stays fp32 bit for bit.  RGB / thermal agree
 *                                 with the fp32 path to ~1e-5; tests/test_head_bf16x3_gpu.py, bench.py extra.head_bf16x3
 *    The Python package reads TN_FUSE_SMALL=0 (one launch per reference seam instead of the fused small kernels: test aid), TN_DM_PREFETCH=0 (the
 *    device data manager launches every batch itself instead of handing the next one to tn_train_step: A/B timing), TN_NEXT_SAMPLING=0 (the engine
 *    plans no next_sampling), TN_DP_SCHEDULE=overlapped|simple (pins the data-parallel exchange schedule of trainer.FusedTrainerMixin) and writes
 *    nothing into the environment (rounds 2-4 set GPU_MAX_HW_QUEUES=8 at import: the schedules now fit the runtime's default of four).
 */
#ifndef THERMAL_NERF_HIP_H
#define THERMAL_NERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported (tests/test_abi_cpu.py compares `nm -D`). */
#if defined(__GNUC__) || defined(__clang__)
#define TN_API __attribute__((visibility("default")))
#else
#define TN_API
#endif

#define TN_OK 0
#define TN_EINVAL (-22)   /* bad argument / unsupported shape */
#define TN_ELAUNCH (-5)   /* hipLaunch / runtime error (see tn_last_error) */
#define TN_MAX_LEVELS 16
#define TN_MAX_SAMPLES 256 /* samples per ray per level (one 64-lane wave x 4) */

typedef void* tn_stream_t;

/* Multiresolution hash grid as the reference's torch HashEncoding lays it out
 * (field_components/encodings.py:343-347,377-379): table [num_levels * 2^log2_hashmap_size, 2]. */
typedef struct TnGrid {
  const float* table;
  float* table_grad; /* may be NULL when no gradient is wanted */
  int32_t num_levels;
  int32_t log2_hashmap_size;
  float res[TN_MAX_LEVELS]; /* floor(min_res * growth^l) as fp32, computed by the host exactly as the reference does */
  /* NULL, or one float on the device that every table-gradient scatter of this grid sets to 1.0f when an entry of table_grad comes out inf / NaN
   * (GradScaler's found_inf for the optimiser group that owns the table, engine/trainer.py:470-495 -> torch/amp/grad_scaler.py: raised by the
   * kernel that writes the final value, so no separate pass over the 64 MB of table gradient is needed; never cleared by the scatter) */
  float* nonfinite_flag;
  /* The caller's PROMISE (not checked): every entry of table_grad is zero when a table-gradient scatter of this grid starts, and this scatter is
   * the only writer of table_grad until it ends.  The fold pass then STORES a slot's sum instead of adding it to what the slot holds (no read of
   * the 64 MB before the write: the main grid's scatter 159 -> 147 us).  A trainer whose optimiser launch clears the gradients behind its read
   * (tn_adam_step_ranges_amp(zero_grads)) and scatters once per table and iteration may set it; gradient accumulation over several backward
   * passes, or two scatters into one table (separate density fields with the density loss), must leave it 0. */
  int32_t table_grad_is_zero;
} TnGrid;

/* HashMLPDensityField (fields/density_fields.py:34-118): 5 lvl x 2 feat -> Linear(10,16) ReLU Linear(16,1). */
typedef struct TnPropNet {
  TnGrid grid;
  const float *w0, *b0, *w1, *b1; /* [16,10] [16] [1,16] [1] as nn.Linear stores them */
  float *gw0, *gb0, *gw1, *gb1;   /* gradient accumulators, may be NULL */
} TnPropNet;

/* ThermalNerfactoField (fields/thermal_nerfacto_field.py:37-99 over fields/nerfacto_field.py:73-202):
 * 16 lvl x 2 feat -> Linear(32,64) ReLU Linear(64,16); head Linear(63,64) ReLU Linear(64,64) ReLU Linear(64,C) sigmoid. */
typedef struct TnField {
  TnGrid grid;
  const float *w0, *b0, *w1, *b1;             /* [64,32] [64] [16,64] [16] */
  const float *hw0, *hb0, *hw1, *hb1, *hw2, *hb2; /* [64,63] [64] [64,64] [64] [C,64] [C] */
  const float* emb;                            /* [num_images, 32] appearance embedding */
  float *gw0, *gb0, *gw1, *gb1, *ghw0, *ghb0, *ghw1, *ghb1, *ghw2, *ghb2, *gemb; /* gradients, may be NULL */
  int32_t num_channels; /* C: 4 (shared RGBT), 3 or 1 (separate) */
  int32_t num_images;
} TnField;

TN_API const char* tn_last_error(void);
TN_API int tn_version(void);
/* bytes of device scratch tn_field_* need for `num_points` samples (packed weights + saved activations). */
TN_API int64_t tn_field_workspace_bytes(int64_t num_points, int32_t training);
/* Work plan of the field's hash-grid gather (HashEncoding.pytorch_fwd, field_components/encodings.py:401-461), for tests and diagnostics.
 * The gather is XCD-affine: workgroup b runs on XCD b % 8 and reads ONE level, so that a hashed level's table is served from that XCD's L2;
 * XCD x runs level x for all samples, then level x + 8.  out [8][2][3] int32: item i of XCD x = {level (-1: unused), first chunk,
 * chunk count}.  Returns the number of chunks per level (every level's chunks appear exactly once in the plan) or TN_EINVAL. */
TN_API int32_t tn_field_encode_plan(const TnGrid* grid, int64_t num_points, int32_t* out);

/* ---- N2  PatchPixelSampler.sample on a jagged image list + ground-truth gather (data/pixel_samplers.py:296-337 collate_image_dataset_batch_list,
 *          :389-441 PatchPixelSampler.sample_method without masks; what VanillaDataManager.next_train does on the host every step,
 *          data/datamanagers/base_datamanager.py:538-547).  The cached training images stay resident in HBM:
 * images: all images back to back, image i is [height[i], width[i], 3] fp32 starting at float offset image_offsets[i];
 * is_thermal [num_images] fp32 by batch position; image_idx [num_images] int64 = dataset (camera) index of each batch position;
 * u [num_rays / patch^2, 3] fp32 uniforms in [0,1) -- exactly what torch.rand returns in the reference, image after image (column 0 unused).
 * Every image gets (num_rays / num_images) / patch^2 patches and the last one the remainder, which must be a whole number of patches
 * (the reference asserts the same).  patch_size 1..8.
 * Outputs: ray_indices [N,3] int64 (camera,row,col), image [N,3], is_thermal_out [N], camera_indices [N] int64 (= ray_indices[:,0] as the
 * contiguous vector the field kernels take; may be NULL). */
TN_API int tn_sample_pixels(const float* images, const int64_t* image_offsets, const int32_t* heights, const int32_t* widths,
                     const float* is_thermal, const int64_t* image_idx, int32_t num_images, const float* u, int64_t num_rays,
                     int32_t patch_size, int64_t* ray_indices, float* image, float* is_thermal_out, int64_t* camera_indices,
                     tn_stream_t stream);

/* ---- a1  RayGenerator.forward -> Cameras._generate_rays_from_coords (model_components/ray_generators.py:40-55,
 *          cameras/cameras.py:598-655,781-786,886-909; undistortion cameras/camera_utils.py:409-446).
 * ray_indices [N,3] int64 (camera,row,col); c2w [C,3,4]; fx,fy,cx,cy [C]; distortion [C,6] (k1,k2,k3,k4,p1,p2) or NULL. */
TN_API int tn_raygen(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy, const float* cx,
              const float* cy, const float* distortion, int32_t num_cameras, int64_t N, float* origins, float* directions,
              float* pixel_area, float* directions_norm, tn_stream_t stream);
/* N2 + a1 in ONE launch (VanillaDataManager.next_train, data/datamanagers/base_datamanager.py:538-547: pixel sampler -> ground truth ->
 * RayGenerator): arguments of tn_sample_pixels followed by those of tn_raygen without ray_indices (the sampled pixel is handed over in
 * registers; ray_indices is still written).  Same results as the two calls. 
 * pixel_area may be NULL (a trainer whose model does not read it: the two extra undistortions per ray that only serve it are skipped). */
/* The arguments of tn_sample_rays as a block: TnTrainStep::next_sample hands the NEXT iteration's batch to tn_train_step, which samples it in
 * co-work blocks of its optimiser launch (the last launch of the iteration: an HBM-bound pass beside a short, latency-bound one). */
typedef struct TnSampleRays {
  const float* images; const int64_t* image_offsets; const int32_t* heights; const int32_t* widths; const float* is_thermal; const int64_t* image_idx;
  int32_t num_images; const float* u; int64_t num_rays; int32_t patch_size;
  int64_t* ray_indices; float* image; float* is_thermal_out; int64_t* camera_indices;
  const float* c2w; const float* fx; const float* fy; const float* cx; const float* cy; const float* distortion; int32_t num_cameras;
  float* origins; float* directions; float* pixel_area; float* directions_norm;
} TnSampleRays;
TN_API int tn_sample_rays_args(const TnSampleRays* args, tn_stream_t stream); /* tn_sample_rays on the block */
TN_API int tn_sample_rays(const float* images, const int64_t* image_offsets, const int32_t* heights, const int32_t* widths,
                   const float* is_thermal, const int64_t* image_idx, int32_t num_images, const float* u, int64_t num_rays,
                   int32_t patch_size, int64_t* ray_indices, float* image, float* is_thermal_out, int64_t* camera_indices,
                   const float* c2w, const float* fx, const float* fy, const float* cx, const float* cy, const float* distortion,
                   int32_t num_cameras, float* origins, float* directions, float* pixel_area, float* directions_norm, tn_stream_t stream);

/* ---- a4  CameraOptimizer(SO3xR3).apply_to_raybundle (cameras/camera_optimizers.py:130-176, cameras/lie_groups.py:24-58).
 * pose_adjustment [C,6]; frozen [C] uint8 (1 = non-trainable camera -> identity); camera_indices [N] int64. */
TN_API int tn_pose_apply_fwd(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* origins_in,
                      const float* directions_in, int64_t N, int32_t num_cameras, float* origins_out, float* directions_out,
                      tn_stream_t stream);
/* backward: d_origins/d_directions [N,3] -> accumulates into grad_pose [C,6]. */
TN_API int tn_pose_apply_bwd(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                      const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose,
                      tn_stream_t stream);

/* ---- a6  UniformLinDispPiecewiseSampler / SpacedSampler.generate_ray_samples (model_components/ray_samplers.py:78-128,225-248).
 * lin_bins [S+1] = torch.linspace(0,1,S+1) supplied by the host; jitter [N] or NULL (eval). Outputs s_bins,e_bins [N,S+1]. */
TN_API int tn_spaced_bins(const float* lin_bins, const float* jitter, const float* nears, const float* fars, int64_t N, int32_t S,
                   float* s_bins, float* e_bins, tn_stream_t stream);
/* tn_pose_apply_fwd and tn_spaced_bins (the two independent first steps of a training render) in one launch; arguments of the former, then of
 * the latter without N. */
TN_API int tn_pose_spaced_bins(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* origins_in,
                        const float* directions_in, int64_t N, int32_t num_cameras, float* origins_out, float* directions_out,
                        const float* lin_bins, const float* jitter, const float* nears, const float* fars, int32_t S, float* s_bins,
                        float* e_bins, tn_stream_t stream);

/* ---- a8/a9/a10  Field.density_fn -> HashMLPDensityField.get_density (fields/base_field.py:48-68, fields/density_fields.py:95-118,
 *          field_components/encodings.py:401-461, field_components/mlp.py:159-178, field_components/activations.py:28-41).
 * positions = origins + directions * (e_bins[s]+e_bins[s+1])/2 (cameras/rays.py:49-58). density [N,S]. */
TN_API int tn_prop_density_fwd(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins, int64_t N,
                        int32_t S, float* density, tn_stream_t stream);
/* backward of the above: d_density [N,S] -> table/MLP gradients (accumulated) and, if non-NULL, d_origins/d_directions [N,3] (accumulated).
 * workspace: tn_prop_workspace_bytes(N*S) bytes of device scratch (d enc for the table scatter + the scatter's records); workspace_bytes = the size
 * of the buffer behind it: a shorter one is refused with TN_EINVAL (every workspace_bytes argument below works the same way -- the scratch of
 * a backward pass is hundreds of MB, and a short buffer would be a device out-of-bounds write the callee could not see). */
TN_API int64_t tn_prop_workspace_bytes(int64_t num_points);
TN_API int tn_prop_density_bwd(const TnPropNet* net, const float* origins, const float* directions, const float* e_bins,
                        const float* d_density, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins,
                        float* d_directions, tn_stream_t stream);

/* ---- a9 (backward)  autograd of HashEncoding.pytorch_fwd (field_components/encodings.py:420-461) with respect to the table (and the sample
 *          position): trilinear scatter-add of g_enc [N*S, ld] (feature 2*level + f) into grid->table_grad; d_origins/d_directions optional.
 *          ld == -1: g_enc is LEVEL-major, [num_levels][N*S][2] (what the main field's backward produces: coalesced reads per level).
 *          Used by tn_prop_density_bwd and tn_field_bwd; exposed because it is the dominant kernel of the training step.
 *          workspace: tn_hash_scatter_workspace_bytes(N*S, num_levels) of 256-byte-aligned device scratch (contents irrelevant), or NULL.
 *          With it the scatter is atomic-free: every contribution is written once as a (slot, value) record into the bucket of its
 *          2^12-slot table slice and the buckets are summed in LDS (coarse levels are pre-merged in registers); NULL adds every
 *          contribution straight into table_grad with global float atomics (same result up to summation order, several times slower). */
TN_API int64_t tn_hash_scatter_workspace_bytes(int64_t num_points, int32_t num_levels);
TN_API int tn_hash_scatter(const TnGrid* grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int32_t ld,
                    int64_t N, int32_t S, float* d_origins, float* d_directions, void* workspace, int64_t workspace_bytes, tn_stream_t stream);

/* ---- a11 RaySamples.get_weights (cameras/rays.py:128-150) and, optionally, DepthRenderer("median") of the same level
 *          (model_components/renderers.py:547-557; used for prop_depth_i, models/nerfacto.py:351-352). median_depth may be NULL. */
TN_API int tn_weights_fwd(const float* e_bins, const float* density, int64_t N, int32_t S, float* weights, float* median_depth,
                   tn_stream_t stream);
TN_API int tn_weights_bwd(const float* e_bins, const float* density, const float* weights, const float* d_weights, int64_t N, int32_t S,
                   float* d_density, tn_stream_t stream);

/* ---- a7  PDFSampler.generate_ray_samples incl. the anneal pow of ProposalNetworkSampler (model_components/ray_samplers.py:276-372,602).
 * u_lin [S+1] = torch.linspace(0, 1-1/(S+1), S+1) supplied by the host (eval adds 1/(2(S+1)), train adds jitter/(S+1)). */
TN_API int tn_pdf_resample(const float* s_bins_prev, const float* weights_prev, int32_t S_prev, float anneal, const float* u_lin,
                    const float* jitter, const float* nears, const float* fars, int64_t N, int32_t S, float* s_bins, float* e_bins,
                    tn_stream_t stream);

/* a11 + a7 back to back, as ProposalNetworkSampler.generate_ray_samples always calls them (ray_samplers.py:593-611): the weights of the
 * previous level (written to weights_prev [N,S_prev]; median_prev [N] optional) and the bins of the next level, one launch, the weights
 * handed over in registers.  Bit-identical to tn_weights_fwd followed by tn_pdf_resample. */
TN_API int tn_weights_resample(const float* e_bins_prev, const float* density_prev, const float* s_bins_prev, int32_t S_prev, float anneal,
                        const float* u_lin, const float* jitter, const float* nears, const float* fars, int64_t N, int32_t S,
                        float* weights_prev, float* median_prev, float* s_bins, float* e_bins, tn_stream_t stream);

/* ---- a12/a13/a14  Field.forward = NerfactoField.get_density + get_outputs with ThermalNerfactoField.mlp_head
 *          (fields/base_field.py:114-133, fields/nerfacto_field.py:205-229,272-348, fields/thermal_nerfacto_field.py:91-99).
 * camera_indices [N] int64; training!=0 uses emb[camera], else mean(emb) (use_average_appearance_embedding=True).
 * Outputs: density [N,S], rgb [N,S,C], optional density_before_activation [N,S] (may be NULL).
 * workspace: device scratch of tn_field_workspace_bytes(N*S, training) bytes; when training!=0 it keeps the activations
 * tn_field_bwd consumes, so the same pointer must be passed to both. */
TN_API int tn_field_pack_weights(const TnField* field, void* workspace, tn_stream_t stream);
TN_API int tn_field_fwd(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins,
                 int64_t N, int32_t S, int32_t training, void* workspace, int64_t workspace_bytes, float* density, float* rgb,
                 float* density_pre, tn_stream_t stream);
/* backward: d_density [N,S], d_rgb [N,S,C] -> all TnField gradients (accumulated); d_origins/d_directions optional (accumulated).
 * d_rgb = NULL: density-only backward of tn_field_density_fwd(training != 0) (see there).
 * The appearance embedding's gradients (the embedding rows and the embedding columns of the head's first weight matrix) are formed from
 * per-camera sums kept in the workspace (head layer 0's bias gradient restricted to the camera; the training forward likewise adds a
 * per-camera vector instead of multiplying the embedding per sample): num_images <= TN_FIELD_MAX_IMAGES, refused with TN_EINVAL beyond.  The forward (training != 0) clears
 * those sums and the backward leaves them cleared: the workspace may come from an uninitialised allocation. */
#define TN_FIELD_MAX_IMAGES 4096
TN_API int tn_field_bwd(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices, const float* e_bins,
                 const float* d_density, const float* d_rgb, int64_t N, int32_t S, void* workspace, int64_t workspace_bytes, float* d_origins,
                 float* d_directions, tn_stream_t stream);
/* The same backward in phases, for data-parallel training: the table gradient of a level range is final as soon as its scatter has run,
 * so the caller can start that range's all-reduce (DDP's bucketed reducer, pipelines/base_pipeline.py:281-283) while the next range is
 * scattered.  phases is a bit set; tn_field_bwd == all three with levels [0, num_levels):
 *   TN_BWD_MLP      MLP backward + weight/bias/embedding gradients (the fused chain; the embedding's two gradients are finished from its
 *                   per-camera sums at the head of the d-position launch, or in a small launch of their own without one) and, when
 *                   d_origins is given, d position from the forward's saved
 *                   d enc / d offset (k_field_dpos); with TN_BWD_FORK_DPOS that second launch goes to a library-owned companion stream
 *                   beside the scatter -- worth it only when other streams are busy anyway (a second active queue costs more than it hides)
 *                   (when TN_BWD_MLP and a TN_BWD_SCATTER over all levels come in ONE call and the scatter takes its segmented path, the pass
 *                   gets no launch at all: it runs in extra blocks of the scatter's bin launch, d_origins / d_directions complete when that is)
 *   TN_BWD_SCATTER  table gradient of levels [level_begin, level_end); needs TN_BWD_MLP done
 *   TN_BWD_JOIN     make `stream` wait for the companion stream; required before d_origins / d_directions or the workspace are used again
 *   TN_BWD_SCATTER_BIN / TN_BWD_SCATTER_FOLD  the same scatter in its two passes: BIN writes the (slot, value) records of ALL levels once
 *                   (+ the d_origins/d_directions contribution); FOLD sums the records of levels [level_begin, level_end) into the table
 *                   gradient.  One BIN, then one FOLD per exchanged level range, costs the same as a single TN_BWD_SCATTER over all levels;
 *                   TN_BWD_SCATTER per range repeats the per-sample work of the bin pass for every range.
 *   TN_BWD_COUNTERS_CLEAN  with TN_BWD_SCATTER (whole grid) or TN_BWD_SCATTER_BIN in a call WITHOUT TN_BWD_MLP: the caller vouches that the
 *                   TN_BWD_MLP phase of this workspace and batch has run since the workspace's last scatter.  That launch leaves the bin
 *                   pass's bucket counters zeroed; without the flag a phase-by-phase backward pays a memset launch for them. */
enum { TN_BWD_MLP = 1, TN_BWD_SCATTER = 2, TN_BWD_JOIN = 4, TN_BWD_SCATTER_BIN = 8, TN_BWD_SCATTER_FOLD = 16, TN_BWD_FORK_DPOS = 32,
       TN_BWD_COUNTERS_CLEAN = 64 };
TN_API int tn_field_bwd_phase(const TnField* field, const float* origins, const float* directions, const int64_t* camera_indices,
                       const float* e_bins, const float* d_density, const float* d_rgb, int64_t N, int32_t S, void* workspace,
                       int64_t workspace_bytes, float* d_origins, float* d_directions, int32_t phases, int32_t level_begin, int32_t level_end,
                       tn_stream_t stream);
/* Data-parallel exchange of the COARSE levels in dense form.  Levels whose (res+1)^3 cells are fewer than the table's slots are accumulated
 * in dense per-cell replicas anyway; their slice of the table gradient is almost all zeros (332 k possible non-zeros in 2.6 M slots for
 * levels 0-4 of the default grid), so a data-parallel run exchanges the per-cell sums (2.65 MB) instead of the table slice (20 MB):
 *   n = tn_field_dense_count(field, N*S, lb, le)      float2 cells of levels [lb, le) if ALL of them are dense levels for this batch size, else 0
 *   tn_field_bwd_scatter_dense(..., lb, le, dense_sum) as TN_BWD_SCATTER of that range, but the per-cell sums go to dense_sum [n,2] (written, not
 *                                                     accumulated) and the table gradient of the range is left untouched
 *   (all-reduce dense_sum)
 *   tn_field_dense_fold(field, N*S, lb, le, dense_sum) hashes the sums into the table gradient (accumulating), as the plain scatter would have. */
TN_API int64_t tn_field_dense_count(const TnField* field, int64_t num_points, int32_t level_begin, int32_t level_end);
TN_API int tn_field_bwd_scatter_dense(const TnField* field, const float* origins, const float* directions, const float* e_bins, int64_t N, int32_t S,
                               void* workspace, int64_t workspace_bytes, float* d_origins, float* d_directions, int32_t level_begin, int32_t level_end,
                               float* dense_sum, tn_stream_t stream);
TN_API int tn_field_dense_fold(const TnField* field, int64_t num_points, int32_t level_begin, int32_t level_end, const float* dense_sum,
                        tn_stream_t stream);
/* density only (cross-evaluation density2 / density2_thermal, models/thermal_nerfacto.py:447-458): get_density without get_outputs.
 * training != 0 keeps what the backward needs in `workspace` (sized with tn_field_workspace_bytes(N*S, 1)); that backward is
 * tn_field_bwd / tn_field_bwd_phase with d_rgb = NULL: the colour head, its three weight gradients and the appearance embedding are skipped. */
TN_API int tn_field_density_fwd(const TnField* field, const float* origins, const float* directions, const float* e_bins, int64_t N, int32_t S,
                         int32_t training, void* workspace, int64_t workspace_bytes, float* density, tn_stream_t stream);

/* ---- a15/a16  RGBRenderer / RGBTRenderer (background "last_sample"), AccumulationRenderer, DepthRenderer median+expected
 *          (model_components/renderers.py:118-133,238-245,292-307,418-425,509,547-576).
 * rgb [N,S,C]; outputs comp [N,C], accumulation [N], depth_median [N], depth_expected [N] (unclipped) and
 * steps_minmax [2] (running min/max of the sample midpoints as ordered uint32 bit patterns; initialise with tn_minmax_init,
 * then call tn_clip_depth to apply the batch-global clip of renderers.py:574). */
TN_API int tn_minmax_init(uint32_t* steps_minmax, tn_stream_t stream);
TN_API int tn_composite_fwd(const float* rgb, const float* weights, const float* e_bins, int64_t N, int32_t S, int32_t C, int32_t training,
                     float* comp, float* accumulation, float* depth_median, float* depth_expected, uint32_t* steps_minmax,
                     tn_stream_t stream);
TN_API int tn_clip_depth(float* depth_expected, const uint32_t* steps_minmax, int64_t N, tn_stream_t stream);
/* backward (train mode): d_comp [N,C] -> d_rgb [N,S,C] (written) and d_weights [N,S] (accumulated). */
TN_API int tn_composite_bwd(const float* rgb, const float* weights, const float* d_comp, int64_t N, int32_t S, int32_t C, float* d_rgb,
                     float* d_weights, tn_stream_t stream);

/* a11 + a15 + a16 of the last sampling level in ONE launch: RaySamples.get_weights (cameras/rays.py:128-150) followed by every renderer above
 * (models/nerfacto.py:330-340), the weights handed over in registers; a second small launch applies the batch-global clip of the expected
 * depth.  Results are bit-identical to tn_weights_fwd + tn_minmax_init + tn_composite_fwd + tn_clip_depth (two launches, not four, and no
 * atomics).  scratch: TN_RENDER_SCRATCH_FLOATS floats of device memory, contents irrelevant (required when depth_expected is given; one
 * buffer per stream that may run this concurrently).  accumulation / depth_median / depth_expected may be NULL. */
#define TN_RENDER_SCRATCH_FLOATS 4096
TN_API int tn_render_fwd(const float* e_bins, const float* density, const float* rgb, int64_t N, int32_t S, int32_t C, int32_t training,
                  float* weights, float* comp, float* accumulation, float* depth_median, float* depth_expected, float* scratch,
                  tn_stream_t stream);
/* its backward: tn_composite_bwd followed by tn_weights_bwd in one launch.  d_weights_in [N,S] = gradient that reaches the weights from the
 * losses (read only: the compositing term is added in registers); d_rgb [N,S,C] and d_density [N,S] are written. */
TN_API int tn_render_bwd(const float* e_bins, const float* density, const float* rgb, const float* weights, const float* d_comp,
                  const float* d_weights_in, int64_t N, int32_t S, int32_t C, float* d_rgb, float* d_density, tn_stream_t stream);

/* ---- a5..a17 in ONE call: the no-grad render of one branch (a sampler with two proposal networks + a field), i.e.
 * ThermalNerfactoModel.get_outputs at inference (models/nerfacto.py:299-353 via models/thermal_nerfacto.py:403-445; ProposalNetworkSampler
 * model_components/ray_samplers.py:577-618 without jitter; mean appearance embedding; "last_sample" background; depth clip).  The library
 * enqueues tn_spaced_bins -> tn_prop_density_fwd -> tn_weights_resample -> tn_prop_density_fwd -> tn_weights_resample ->
 * tn_field_pack_weights + tn_field_fwd -> tn_render_fwd itself: same kernels, same results as those calls made one by one.
 * nears / fars [N]; lin_spaced0 [S0+1] = linspace(0,1,S0+1), lin_pdf_k [S_k+1] = linspace(0, 1 - 1/(S_k+1), S_k+1) as the reference builds
 * them on the host; anneal = the sampler's current histogram-padding exponent.  Outputs: rgb [N,C], density [N,S2] (required);
 * accumulation, depth_median, depth_expected, prop_depth0/1 [N], e_bins_out [N,S2+1], rgb_samples_out [N,S2,C] (each may be NULL).
 * workspace: tn_render_rays_eval_workspace_bytes(N, S0, S1, S2, C) bytes, 256-byte aligned. */
TN_API int64_t tn_render_rays_eval_workspace_bytes(int64_t num_rays, int32_t S0, int32_t S1, int32_t S2, int32_t C);
TN_API int tn_render_rays_eval(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* origins,
                        const float* directions, const int64_t* camera_indices, const float* nears, const float* fars, int64_t N,
                        int32_t S0, int32_t S1, int32_t S2, float anneal, const float* lin_spaced0, const float* lin_pdf1,
                        const float* lin_pdf2, void* workspace, int64_t workspace_bytes, float* rgb, float* accumulation, float* depth_median,
                        float* depth_expected, float* prop_depth0, float* prop_depth1, float* density, float* e_bins_out,
                        float* rgb_samples_out, tn_stream_t stream);

/* The same for TRAINING: one call = CameraOptimizer.apply_to_raybundle (pose_adjustment may be NULL) + ProposalNetworkSampler with jitter
 * (jitter_k [N] or NULL per level) + Field.forward with the activations kept in field_workspace (tn_field_workspace_bytes(N*S2, 1)) for
 * tn_field_bwd + get_weights + renderers.  Launches: tn_pose_spaced_bins, tn_prop_density_fwd, tn_weights_resample, tn_prop_density_fwd,
 * tn_weights_resample, tn_field_pack_weights, tn_field_fwd(training), tn_render_fwd(training): results identical to those calls.
 * Everything later stages need goes into ONE buffer `out` (256-byte aligned); tn_render_rays_train_layout fills offsets[] (floats) with the
 * position of: 0 origins 1 directions (pose-corrected; unused when pose_adjustment is NULL) | 2 s_bins0 3 e_bins0 4 density0 5 weights0
 * 6 median0 | 7..11 the same for level 1 | 12 s_bins2 13 e_bins2 14 density2 15 weights2 | 16 rgb_samples [N,S2,C] 17 comp [N,C]
 * 18 accumulation 19 depth_median 20 depth_expected [N] | 21 scratch | 22, 23 the proposal levels' encodings, N*S0*10 and N*S1*10 floats
 * (level-major [5][N*S_k] float2; written only with save_prop_enc) | 24 = total floats; num_offsets >= TN_RENDER_TRAIN_OFFSETS.
 * save_prop_enc != 0: the proposal networks take a gradient this iteration (ray_samplers.py:591): their encodings are kept in `out`, and
 * tn_render_rays_train_bwd(prop_enc_saved != 0) reads them back instead of repeating the 40 table reads per sample -- the same values.
 * wait_event_before_field: NULL, or a hipEvent_t that `stream` waits for right before the field's first read of its parameters -- a trainer
 * that runs the previous iteration's Adam launch over the field on another stream lets it overlap the proposal sampling this way.
 * zero_fill: NULL, or zero_fill_bytes (multiple of 16) of 16-byte aligned device memory that the call clears on its way (inside the field's
 * first launch): the caller's zero-initialised accumulators of the iteration -- loss sums, d(composite), d(weights), d origins / d directions --
 * without a fill launch of their own. */
#define TN_RENDER_TRAIN_OFFSETS 25
TN_API int tn_render_rays_train_layout(int64_t num_rays, int32_t S0, int32_t S1, int32_t S2, int32_t C, int64_t* offsets, int32_t num_offsets);
TN_API int tn_render_rays_train(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* pose_adjustment,
                         const uint8_t* frozen, int32_t num_cameras, const float* origins_in, const float* directions_in,
                         const int64_t* camera_indices, const float* nears, const float* fars, int64_t N, int32_t S0, int32_t S1,
                         int32_t S2, float anneal, const float* jitter0, const float* jitter1, const float* jitter2,
                         const float* lin_spaced0, const float* lin_pdf1, const float* lin_pdf2, void* field_workspace,
                         int64_t field_workspace_bytes, float* out, void* wait_event_before_field, void* zero_fill, int64_t zero_fill_bytes,
                         int32_t save_prop_enc, tn_stream_t stream);

/* The TRAINING backward of one branch as ONE call, the counterpart of tn_render_rays_train: everything autograd runs behind d(composite) and
 * d(weights) in ThermalNerfactoModel's training step (models/thermal_nerfacto.py:403-489 backwards; cameras/rays.py:128-150,
 * model_components/renderers.py, fields/nerfacto_field.py:205-348, fields/density_fields.py:95-118 under autograd).  fwd_out is the buffer
 * tn_render_rays_train filled (same N, S, field); origins / directions [N,3] the pose-corrected rays that forward used; d_comp [N,C]; d_weights2 [N,S2] (losses on the fine weights); d_weights0 / d_weights1
 * [N,S0] / [N,S1] or both NULL when the proposal networks take no gradient this iteration (model_components/ray_samplers.py:591,605-610);
 * d_density_extra [N,S2] or NULL (the density loss's gradient on this branch's density, separate mode).  The library enqueues tn_render_bwd,
 * then -- on two companion streams of `stream` -- tn_weights_bwd + tn_prop_density_bwd per proposal level, beside tn_field_bwd on `stream`, and
 * joins: same launches and results as those calls made one by one.  tmp: tn_render_rays_train_bwd_tmp_floats(...) floats of scratch;
 * prop_workspace_k: tn_prop_workspace_bytes(N*S_k) (may be NULL without d_weights).  d_origins / d_directions [N,3] accumulate (or both NULL).
 * prop_enc_saved != 0: fwd_out holds the proposal levels' encodings (tn_render_rays_train(save_prop_enc != 0) on this buffer). */
TN_API int64_t tn_render_rays_train_bwd_tmp_floats(int64_t num_rays, int32_t S0, int32_t S1, int32_t S2, int32_t C);
TN_API int tn_render_rays_train_bwd(const TnPropNet* prop0, const TnPropNet* prop1, const TnField* field, const float* origins,
                             const float* directions, const int64_t* camera_indices, int64_t N, int32_t S0, int32_t S1, int32_t S2,
                             const float* fwd_out, const float* d_comp, const float* d_weights0,
                             const float* d_weights1, const float* d_weights2, const float* d_density_extra, void* field_workspace,
                             int64_t field_workspace_bytes, void* prop_workspace0, int64_t prop_workspace_bytes0, void* prop_workspace1,
                             int64_t prop_workspace_bytes1, float* tmp, float* d_origins, float* d_directions, int32_t prop_enc_saved,
                             tn_stream_t stream);

/* ---- a18  interlevel_loss / distortion_loss (model_components/losses.py:57-158), forward value + gradient in one pass.
 * loss_out[0] += mult * mean_over_rays(...); d_weights accumulated (may be NULL to skip the gradient). */
TN_API int tn_distortion_loss(const float* s_bins, const float* weights, int64_t N, int32_t S, float mult, float* loss_out, float* d_weights,
                       tn_stream_t stream);
TN_API int tn_interlevel_loss(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, const float* s_bins_prop,
                       const float* weights_prop, int32_t S_prop, int64_t N, float mult, float* loss_out, float* d_weights_prop,
                       tn_stream_t stream);

/* Both of the above for one branch in ONE launch (the "K7" entry point of SURVEY 8b): distortion on the fine level + interlevel against each of
 * the num_props (<= TN_MAX_PROP_LEVELS) proposal levels.  s_bins_prop / weights_prop / S_prop / d_weights_prop are HOST arrays of num_props
 * entries (device pointers / sizes); d_weights_prop[i] and d_weights_fine may be NULL.  Same accumulation semantics as the single calls. */
#define TN_MAX_PROP_LEVELS 4
TN_API int tn_proposal_losses(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, int32_t num_props,
                       const float* const* s_bins_prop, const float* const* weights_prop, const int32_t* S_prop,
                       float* const* d_weights_prop, int64_t N, float distortion_mult, float interlevel_mult, float* distortion_out,
                       float* interlevel_out, float* d_weights_fine, tn_stream_t stream);

/* ---- a19  ThermalNerfactoModel.get_loss_dict pixel terms (models/thermal_nerfacto.py:284-354, model_components/losses.py:602-651,
 *          utils/rgbt_utils.py:6-32): rgb MSE, thermal MSE x thermal_mult, 2x2-patch TV, cross-channel gradient loss.
 * pred_rgb [N,3], pred_thermal [N,1] (for shared mode both are views of one [N,4] buffer: pass strides in floats),
 * image [N,3], is_thermal [N] float.  losses_out[0..3] += {rgb, thermal, tv_pixel, cross_channel}; losses_out must have room for 8 floats
 * (entry 4 is scratch: the number of RGB rays); d_pred_* accumulated. */
TN_API int tn_pixel_losses(const float* pred_rgb, int32_t rgb_stride, const float* pred_thermal, int32_t thermal_stride, const float* image,
                    const float* is_thermal, int64_t N, float thermal_mult, float tv_mult, float cross_mult, float* losses_out,
                    float* d_pred_rgb, float* d_pred_thermal, tn_stream_t stream);
/* The proposal losses of one branch AND (pred_rgb != NULL) the pixel terms above in ONE launch: they are independent, and each alone is a
 * short latency-bound kernel (get_loss_dict's distortion / interlevel / rgb / thermal / tv / cross-channel terms of
 * models/thermal_nerfacto.py:284-368 side by side).  Arguments as tn_proposal_losses and tn_pixel_losses, except for where the sums go:
 * every block of every term would end with a float atomic into the same 64-byte line, and those execute one after the other (~25 ns each:
 * 40 us for 1500 blocks), so the sums are spread over loss_lines [TN_LOSS_LINES][16] (zero-filled by the caller; block b adds into line
 * b % TN_LOSS_LINES; slots 0 rgb, 1 thermal, 2 tv_pixel, 3 cross_channel, 4 / 5 ray counts per spectrum, 8 interlevel, 9 distortion) and
 * tn_losses_finish adds the lines up: losses16[k] += sum over lines of loss_lines[.][k].  tn_losses_finish also evaluates the camera
 * regulariser of one pose tensor in the same single-block launch when pose_adjustment != NULL (arguments of tn_camera_reg; reg_out may be
 * one of the 16 slots). */
#define TN_LOSS_LINES 64
TN_API int tn_train_losses(const float* s_bins_fine, const float* weights_fine, int32_t S_fine, int32_t num_props,
                    const float* const* s_bins_prop, const float* const* weights_prop, const int32_t* S_prop,
                    float* const* d_weights_prop, int64_t N, float distortion_mult, float interlevel_mult, float* d_weights_fine,
                    const float* pred_rgb, int32_t rgb_stride, const float* pred_thermal, int32_t thermal_stride, const float* image,
                    const float* is_thermal, float thermal_mult, float tv_mult, float cross_mult, float* d_pred_rgb, float* d_pred_thermal,
                    float* loss_lines, tn_stream_t stream);
/* tn_render_fwd(training) + tn_train_losses + tn_render_bwd of the shared-density model's last level in ONE launch (the renderers of
 * model_components/renderers.py:118-133,547-576, the loss terms of models/thermal_nerfacto.py:284-368 + model_components/losses.py:57-158 and the
 * renderers' backward): three wave-per-ray kernels over the same rays, each latency-bound alone.  A block renders one 2x2 patch (4 rays), hands
 * the four composites around in LDS for the pixel terms, adds the distortion term and runs the renderers' backward with the weights still in
 * registers; the interlevel terms run beside it in blocks of their own (they recompute the fine weights with the same instructions).
 * Arguments as the three entry points: C must be 4 (RGB + thermal composite; pred_rgb = comp, pred_thermal = comp + 3, d_comp [N,4] likewise),
 * N a multiple of 4; d_weights_fine [N,S] and d_comp [N,4] are accumulators (read, term added, written), d_weights_prop[i] too (or NULL);
 * d_rgb [N,S,4] and d_density [N,S] are written.  clip_depth = 0 leaves depth_expected unclipped and the per-block min / max of the sample
 * midpoints in `scratch` (tn_render_fwd's clip launch is then the caller's to issue).  Every per-element result equals the three calls' bit for
 * bit; the loss sums (loss_lines) are added up in another order, so they agree to rounding -- as two runs of tn_train_losses do. */
TN_API int tn_render_losses_bwd(const float* e_bins, const float* density, const float* rgb, int64_t N, int32_t S, int32_t C, float* weights,
                         float* comp, float* accumulation, float* depth_median, float* depth_expected, float* scratch,
                         const float* s_bins_fine, int32_t num_props, const float* const* s_bins_prop, const float* const* weights_prop,
                         const int32_t* S_prop, float* const* d_weights_prop, float distortion_mult, float interlevel_mult,
                         float* d_weights_fine, const float* image, const float* is_thermal, float thermal_mult, float tv_mult, float cross_mult,
                         float* d_comp, float* loss_lines, float* d_rgb, float* d_density, int32_t clip_depth, tn_stream_t stream);
TN_API int tn_losses_finish(const float* loss_lines, float* losses16, const float* pose_adjustment, int32_t num_cameras, float trans_pen,
                     float rot_pen, float scale, float* reg_out, float* grad_pose, tn_stream_t stream);
/* tn_pose_apply_bwd + tn_losses_finish for the same pose tensor in one launch (the end of an iteration's backward); loss_lines / losses16 may
 * both be NULL (regulariser only). */
TN_API int tn_pose_bwd_finish(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                       const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose,
                       const float* loss_lines, float* losses16, float trans_pen, float rot_pen, float scale, float* reg_out,
                       tn_stream_t stream);
/* tn_pose_bwd_finish + GradScaler's non-finite check of what the table scatters do not see, in the same launch: found_inf[pose_flag] is raised
 * when a contribution to the pose gradient is inf / NaN, and found_inf[flag_index[k]] when any of the `counts[k]` gradients at grads + offsets[k]
 * is (up to 8 SMALL ranges -- MLP weights, embeddings: at most 4 M floats each; offsets / counts / flag_index are HOST arrays).  With
 * TnGrid::nonfinite_flag on every grid this replaces tn_grad_nonfinite_ranges over the whole gradient arena. */
TN_API int tn_pose_bwd_finish_check(const float* pose_adjustment, const uint8_t* frozen, const int64_t* camera_indices, const float* directions_in,
                             const float* d_origins, const float* d_directions, int64_t N, int32_t num_cameras, float* grad_pose,
                             const float* loss_lines, float* losses16, float trans_pen, float rot_pen, float scale, float* reg_out,
                             const float* grads, int32_t num_ranges, const int64_t* offsets, const int64_t* counts, const int32_t* flag_index,
                             int32_t num_flags, float* found_inf, int32_t pose_flag, tn_stream_t stream);
/* density L1 cross loss with the reference's detach asymmetry (models/thermal_nerfacto.py:328-344): loss += a*mean|x-y| with
 * gradient weight gx to x and gy to y (accumulated; either may be NULL). */
TN_API int tn_l1_loss(const float* x, const float* y, int64_t count, float gx, float gy, float* loss_out, float* d_x, float* d_y,
               tn_stream_t stream);
/* per-iteration metrics in one launch (models/thermal_nerfacto.py:262-270 PSNR per spectrum from the tn_pixel_losses sums losses[0,1,4,5];
 * cameras/camera_optimizers.py:197-202 pose norms): metrics_out[0] psnr_rgb, [1] psnr_thermal, [2],[3] |pose0[:, :3]|, |pose0[:, 3:]|,
 * [4],[5] the same for pose1; either pose may be NULL. */
TN_API int tn_train_metrics(const float* losses, int64_t N, float thermal_mult, const float* pose0, int32_t num_cameras0, const float* pose1,
                     int32_t num_cameras1, float* metrics_out, tn_stream_t stream);
/* camera regulariser (cameras/camera_optimizers.py:189-195). */
TN_API int tn_camera_reg(const float* pose_adjustment, int32_t num_cameras, float trans_pen, float rot_pen, float scale, float* loss_out,
                  float* grad_pose, tn_stream_t stream);

/* ---- N1  torch.optim.Adam(lr, eps) as engine/optimizers.py:73-210 builds it, fused over a flat fp32 arena.
 * step is 1-based. */
TN_API int tn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, int32_t step, double lr,
                 double beta1, double beta2, double eps, tn_stream_t stream);
/* The same for several ranges of one set of arenas in ONE launch: range k covers elements [offsets[k], offsets[k] + counts[k]) (offsets
 * multiples of 4) with its own 1-based step count and learning rate -- the optimiser groups of engine/optimizers.py:86-112.  offsets, counts,
 * steps, lrs are HOST arrays of num_ranges (<= 8) entries. */
TN_API int tn_adam_step_ranges(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                        const int64_t* counts, const int32_t* steps, const double* lrs, double beta1, double beta2, double eps,
                        tn_stream_t stream);
/* GradScaler semantics without a host round trip (engine/trainer.py:470-495 -> torch/amp/grad_scaler.py; engine/optimizers.py:144-183).
 * GradScaler decides per optimiser, i.e. per parameter group: found_inf is a device array of num_flags floats, one per group.
 *  - tn_grad_nonfinite sets *found_inf = 1 when any of `count` gradients is inf or NaN (it never clears it: zero-fill once per step).
 *  - tn_adam_step_ranges_amp is tn_adam_step_ranges with the decision on the device.  Range k belongs to group flag_index[k] (HOST array, NULL =
 *    all 0):  found_inf[flag] != 0 -> range k is not touched (parameters and both moments bit-identical) and, when count_skip != 0,
 *    skipped[flag] += 1 (once per flag and launch, however many ranges carry it);  inv_scale (device float or NULL): gradients are multiplied by *inv_scale as they are read;  skipped (device int32
 *    array or NULL): the bias corrections use steps[k] - skipped[flag], as torch's fused Adam keeps its step tensors (torch/optim/adam.py:
 *    step -= found_inf);  lr_finals / sched_max_steps (HOST arrays or NULL) + sched_step: when given, lrs[k] is lr_init and the kernel
 *    evaluates the reference's ExponentialDecayScheduler (engine/schedulers.py:109-141) at sched_step - skipped[lag_index] (lag_index = -1:
 *    no lag): the trainer does not step the schedulers in an iteration whose scale dropped (engine/trainer.py:491-495).
 *    zero_grads != 0: the launch CONSUMES the gradients of its ranges -- sets them to zero behind the read, also when the step is skipped --
 *    (`grads` is then written): the optimizers.zero_grad_some() of the next iteration (engine/trainer.py:463-467) without a fill launch.
 *  - tn_grad_scaler_update is GradScaler.update() on the device (backoff / growth of *scale, growth tracker) and adds 1 to *lag (may be NULL)
 *    when any of the num_flags entries of found_inf is set; clear_found_inf != 0 zero-fills found_inf afterwards (ready for the next step). */
TN_API int tn_grad_nonfinite(const float* grads, int64_t count, float* found_inf, tn_stream_t stream);
/* the same for up to 8 ranges of one gradient arena in ONE launch: range k raises found_inf[flag_index[k]] (offsets, counts, flag_index: HOST arrays) */
TN_API int tn_grad_nonfinite_ranges(const float* grads, int32_t num_ranges, const int64_t* offsets, const int64_t* counts, const int32_t* flag_index,
                             int32_t num_flags, float* found_inf, tn_stream_t stream);
TN_API int tn_adam_step_ranges_amp(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                            const int64_t* counts, const int32_t* steps, const double* lrs, const double* lr_finals,
                            const int32_t* sched_max_steps, int32_t sched_step, double beta1, double beta2, double eps,
                            const float* inv_scale, const float* found_inf, const int32_t* flag_index, int32_t num_flags, int32_t* skipped,
                            int32_t lag_index, int32_t count_skip, int32_t zero_grads, tn_stream_t stream);
TN_API int tn_grad_scaler_update(float* scale, int32_t* growth_tracker, float* found_inf, int32_t num_flags, int32_t* lag,
                          double growth_factor, double backoff_factor, int32_t growth_interval, int32_t clear_found_inf, tn_stream_t stream);
/* tn_adam_step_ranges_amp + tn_grad_scaler_update(clear_found_inf = 1) in ONE launch: the last block of the Adam launch to finish performs
 * GradScaler.update() -- every block has read found_inf / the schedule lag by then.  done_counter: TN_ADAM_DONE_WORDS zeroed uint32 on the device
 * (left zero; the blocks count themselves in on 64 counters in 64 different 64-byte lines: same-line atomics execute one after the other). */
#define TN_ADAM_DONE_WORDS (65 * 16)
TN_API int tn_adam_step_ranges_amp_update(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t num_ranges, const int64_t* offsets,
                                   const int64_t* counts, const int32_t* steps, const double* lrs, const double* lr_finals,
                                   const int32_t* sched_max_steps, int32_t sched_step, double beta1, double beta2, double eps,
                                   const float* inv_scale, float* found_inf, const int32_t* flag_index, int32_t num_flags, int32_t* skipped,
                                   int32_t lag_index, int32_t count_skip, int32_t zero_grads, float* scale, int32_t* growth_tracker,
                                   uint32_t* done_counter, double growth_factor, double backoff_factor, int32_t growth_interval,
                                   tn_stream_t stream);
TN_API int tn_fill_zero(void* ptr, int64_t bytes, tn_stream_t stream);

/* ---- Trainer.train_iteration (engine/trainer.py:455-499) for the shared-density model with a camera optimiser, as ONE call: what
 * RenderEngine.train_step enqueues through five calls of this ABI, in the same order on the same streams --
 *   tn_render_rays_train      forward of the branch (pose correction, proposal sampling, field, renderers); the accumulators in `acc` are cleared
 *                             inside the field's first launch
 *   tn_train_losses           pixel terms + distortion + both interlevel terms (value and gradient), sums spread over loss_lines
 *   tn_render_rays_train_bwd  renderer backward, field backward (d position, table scatter), both proposal networks when prop_grad != 0
 *                             (with TN_FUSE_RENDER=1 the renderers at the end of the first, the losses and the renderer backward at the head of the
 *                             third are ONE launch, tn_render_losses_bwd: every gradient and output bit for bit either way, the loss sums to
 *                             rounding)
 *   tn_pose_bwd_finish_check  pose gradient + camera regulariser + loss sums into losses16, GradScaler's found_inf for the pose and the small
 *                             ranges no table scatter sees
 *   tn_adam_step_ranges_amp_update   Adam over the stepped groups (skip / bias correction / LR schedule decided on the device), GradScaler.update()
 * A host that drives training through this entry point spends one call (~0.11 ms of launch time) per iteration; through the Python binding the
 * host side of an iteration drops from 0.39 to 0.22 ms (scripts/trace_fused_host.py, 1024 rays).  The iteration itself is GPU-bound at either
 * batch size (0.44 ms at 1024 rays, 0.83 ms at 4096): the wall time does not change, the host thread is free for the other 0.2 ms.
 * Every pointer is a device pointer unless it says HOST; buffers as the five entry points document them.
 * `acc` (acc_bytes, 16-byte aligned, multiple of 16): ONE allocation holding losses16 [16], loss_lines [TN_LOSS_LINES][16], d_comp [N,C],
 * d_weights2 [N,S2], d_weights0 [N,S0], d_weights1 [N,S1] (both only read when prop_grad != 0), d_origins, d_directions [N,3]; the call clears it. */
#define TN_TRAIN_STEP_MAX_RANGES 8
/* The NEXT iteration's sampling front -- everything tn_render_rays_train runs before the field: CameraOptimizer.apply_to_raybundle
 * (cameras/camera_optimizers.py:130-176), the level-0 bins and twice density_fn -> get_weights -> PDFSampler (model_components/ray_samplers.py:577-618)
 * -- for the batch TnTrainStep::next_sample describes, as co-work of THIS iteration's optimiser launch: five short launches (86 us at 4096 rays) that
 * are bound by instruction issue leave the head of every iteration and run beside the Adam pass over the field, which is bound by HBM and touches
 * none of their inputs.  One wave takes one ray through the whole chain (no ray depends on another); the optimiser groups the chain READS (the
 * proposal networks, the pose corrections: whatever range of `params` holds prop0 / prop1 / pose_adjustment) are stepped by a launch of their own
 * in front, so it sees this iteration's final parameters.  Results land in fwd_out exactly where tn_render_rays_train puts them (origins ... e_bins2,
 * and the proposal encodings when prop_grad != 0), bit for bit.  Uses this call's nears / fars / lin_* tables / pose / frozen / networks and N rays.
 * Taken only when next_sample is, next_sample->num_rays == N, 128 < S0 <= 256, 64 < S1 <= 128 (the sampler's default lane layouts) and at least one
 * stepped range is not read by the chain. */
typedef struct TnNextSampling {
  float* fwd_out;                                                   /* the NEXT iteration's forward buffer (same layout: same N, S0, S1, S2) */
  const float* jitter0; const float* jitter1; const float* jitter2; /* the NEXT iteration's jitter [N] (or NULL) */
  float anneal;                                                     /* the NEXT iteration's histogram-padding exponent */
  int32_t prop_grad;                                                /* the NEXT iteration's prop_grad: != 0 keeps the proposal encodings */
} TnNextSampling;
typedef struct TnTrainStep {
  const TnPropNet* prop0; const TnPropNet* prop1; const TnField* field;  /* HOST structs, gradient pointers set */
  /* the iteration's batch (datamanager.next_train): rays before the pose correction, ground truth */
  const float* origins_in; const float* directions_in; const int64_t* camera_indices; const float* image; const float* is_thermal;
  const float* nears; const float* fars;
  int64_t N; int32_t S0, S1, S2;
  /* camera optimiser */
  const float* pose_adjustment; const uint8_t* frozen; int32_t num_cameras; float* grad_pose;
  float trans_pen, rot_pen, pen_scale;
  /* sampler */
  float anneal; int32_t prop_grad;
  const float* jitter0; const float* jitter1; const float* jitter2; const float* lin_spaced0; const float* lin_pdf1; const float* lin_pdf2;
  /* buffers */
  void* field_workspace; int64_t field_workspace_bytes;
  void* prop_workspace0; int64_t prop_workspace_bytes0; void* prop_workspace1; int64_t prop_workspace_bytes1;
  float* fwd_out;   /* tn_render_rays_train_layout: offsets[TN_RENDER_TRAIN_OFFSETS - 1] floats, 256-byte aligned */
  float* bwd_tmp;   /* tn_render_rays_train_bwd_tmp_floats */
  void* acc; int64_t acc_bytes;
  float* losses16; float* loss_lines; float* d_comp; float* d_weights0; float* d_weights1; float* d_weights2; float* d_origins; float* d_directions;
  /* loss multipliers (models/thermal_nerfacto.py:32-64, models/nerfacto.py:52-133) */
  float thermal_mult, tv_mult, cross_mult, distortion_mult, interlevel_mult;
  /* GradScaler's check of the small gradient ranges (tn_pose_bwd_finish_check) */
  int32_t num_check; int64_t check_offsets[TN_TRAIN_STEP_MAX_RANGES]; int64_t check_counts[TN_TRAIN_STEP_MAX_RANGES];
  int32_t check_flags[TN_TRAIN_STEP_MAX_RANGES]; int32_t pose_flag;
  /* Adam + GradScaler.update (tn_adam_step_ranges_amp_update); num_ranges == 0: no optimiser launch */
  float* params; float* grads; float* exp_avg; float* exp_avg_sq;
  int32_t num_ranges; int64_t offsets[TN_TRAIN_STEP_MAX_RANGES]; int64_t counts[TN_TRAIN_STEP_MAX_RANGES]; int32_t steps[TN_TRAIN_STEP_MAX_RANGES];
  double lrs[TN_TRAIN_STEP_MAX_RANGES]; double lr_finals[TN_TRAIN_STEP_MAX_RANGES]; int32_t sched_max_steps[TN_TRAIN_STEP_MAX_RANGES];
  int32_t flag_index[TN_TRAIN_STEP_MAX_RANGES]; int32_t sched_step;
  double beta1, beta2, eps;
  float* found_inf; int32_t num_flags; int32_t* skipped; int32_t lag_index;
  float* scale; int32_t* growth_tracker; uint32_t* done_counter; double growth_factor, backoff_factor; int32_t growth_interval;
  /* NULL, or the NEXT iteration's batch (HOST struct): sampled in co-work blocks of the optimiser launch; ignored (the caller's to launch,
   * tn_sample_rays_args) when num_ranges == 0 -- *next_sample_taken (HOST, may be NULL) says which */
  const TnSampleRays* next_sample; int32_t* next_sample_taken;
  /* NULL, or (with next_sample) the NEXT iteration's sampling front (HOST struct, see TnNextSampling): run for that batch in the same co-work blocks.
   * *next_sampling_taken (HOST, may be NULL) = 1 when the optimiser launch carried it: the next call then passes sampling_done = 1. */
  const struct TnNextSampling* next_sampling; int32_t* next_sampling_taken;
  /* != 0: the previous call's next_sampling has filled fwd_out up to the field's bins for THIS batch (origins_in / directions_in / camera_indices are
   * next_sample's outputs), with this call's jitter, anneal and prop_grad, and no parameter of the proposal networks or the poses has changed
   * since: the forward starts at the field. */
  int32_t sampling_done;
} TnTrainStep;
TN_API int tn_train_step(const TnTrainStep* step, tn_stream_t stream);
/* ---- 8e  the data-parallel gradient exchange for a host that binds only this ABI: what torch DDP does for the reference
 * (pipelines/base_pipeline.py:281-283: mean all-reduce of the gradients over NCCL; scripts/train.py:138-151 starts one process per GPU).
 * RCCL is resolved at the first call -- from the RCCL the process has loaded already (PyTorch's own), else librccl.so on the loader path -- and is
 * not a link-time dependency of the library; without one these calls return TN_ELAUNCH.  (The Python package exchanges through torch.distributed,
 * nerfstudio_thermal_amd/parallel.py: the same library underneath.)
 *   tn_comm_unique_id(out)            rank 0: 128 bytes (HOST) to hand to every rank (ncclGetUniqueId)
 *   tn_comm_create(id, world, rank, &comm)   every rank, collectively, on its CURRENT device (ncclCommInitRank)
 *   tn_allreduce_grads(comm, grads, count, average, stream)   in place over `count` floats of the caller's gradient arena, enqueued on `stream`:
 *                                     average != 0 -> the mean over the ranks (ncclAvg), else the sum.  Any slice of the arena, as often as the
 *                                     caller's schedule wants (the whole live range behind the backward, or level ranges behind their folds).
 *   tn_comm_destroy(comm) */
TN_API int tn_comm_unique_id(void* unique_id_out);
TN_API int tn_comm_create(const void* unique_id, int32_t world_size, int32_t rank, void** comm_out);
TN_API int tn_comm_destroy(void* comm);
TN_API int tn_allreduce_grads(void* comm, float* grads, int64_t count, int32_t average, tn_stream_t stream);

/* waits for and destroys the library's companion streams (see "State and environment" at the top); 0 or TN_ELAUNCH */
TN_API int tn_shutdown(void);

/* ---------------------------------------------------------------------------------------------------------------------------
 * N4: forward Gaussian-splat render, RGB + thermal colour per Gaussian.  Replaces the gsplat calls of
 * SplatfactoModel.get_outputs (nerfstudio/models/splatfacto.py:739-807): project_gaussians, spherical_harmonics, rasterize_gaussians
 * (colour and depth).  gsplat is a third-party package outside the reference tree: parity is UNPINNED (oracle/splat_oracle.py restates
 * its published algorithm).  16x16 tiles (splatfacto.py:738). */
typedef struct TnSplatCamera {
  float viewmat[12];  /* world -> camera, rows of the 3x4 matrix in gsplat's convention (x right, y down, z forward): splatfacto.py:700-712 */
  float projmat[16];  /* projection_matrix(0.001, 1000, fovx, fovy) @ viewmat, row-major 4x4: splatfacto.py:718,745 */
  float fx, fy, cx, cy;
  float position[3];  /* camera centre in world space (view directions of the SH colours, splatfacto.py:770) */
  float clip_thresh;  /* near clip in view space (gsplat default 0.01) */
  int32_t width, height;
} TnSplatCamera;
/* scratch for num_gaussians Gaussians and up to max_intersections (Gaussian, tile) pairs; num_tiles = ceil(W/16) * ceil(H/16) */
TN_API int64_t tn_splat_workspace_bytes(int64_t num_gaussians, int64_t max_intersections, int32_t num_tiles);
/* project_gaussians + spherical_harmonics (splatfacto.py:739-777).  means [N,3], log_scales [N,3] (exponentiated inside), quats [N,4]
 * (w,x,y,z; normalised inside), opacities [N] (logits), features_dc [N,3], features_rest [N,K,3], thermal_dc [N,1], thermal_rest [N,K,1]
 * (K = num_rest_coeffs).  sh_degree 0..3 = degree evaluated this step (min(step // interval, sh_degree)); -1 = sigmoid(features_dc)
 * (config.sh_degree == 0).  antialiased != 0: opacity x compensation (rasterize_mode "antialiased").  Outputs as gsplat returns them:
 * xys [N,2], depths [N], radii [N] int32, conics [N,3], compensation [N], num_tiles_hit [N] int32, plus tile_box [N,4] int32
 * (x0, y0, x1, y1 in tiles).  Colours and opacities go into the workspace for tn_splat_raster. */
TN_API int tn_splat_project(const TnSplatCamera* camera, const float* means, const float* log_scales, const float* quats, const float* opacities,
                     const float* features_dc, const float* features_rest, const float* thermal_dc, const float* thermal_rest,
                     int64_t num_gaussians, int32_t num_rest_coeffs, int32_t sh_degree, int32_t antialiased, float* xys, float* depths,
                     int32_t* radii, float* conics, float* compensation, int32_t* num_tiles_hit, int32_t* tile_box, void* workspace,
                     int64_t max_intersections, tn_stream_t stream);
/* tile binning of rasterize_gaussians: depth sort of the Gaussians, scan, (tile, Gaussian) pairs in depth order, stable radix sort by
 * tile, tile ranges.  The pairs come from the TIGHT tile boxes tn_splat_project left in the workspace (gsplat's 3-sigma box cut down to
 * the tiles where alpha >= 1/255 is reachable: exact, fewer pairs than sum(num_tiles_hit)).  Reads the pair count back to the host
 * (*num_intersections_out, a HOST pointer; one stream synchronisation, as gsplat's binning does); returns TN_EINVAL with the needed count
 * in *num_intersections_out when it exceeds max_intersections. */
TN_API int tn_splat_bin(const TnSplatCamera* camera, const float* depths, int64_t num_gaussians, void* workspace, int64_t max_intersections,
                 int64_t* num_intersections_out, tn_stream_t stream);
/* rasterize_gaussians, colour (RGB + thermal over background4) and depth in one pass (splatfacto.py:789-809): out_rgbt [H,W,4] clamped to
 * <= 1, out_depth [H,W] = depth / alpha where alpha > 0, else the maximum of the un-normalised depth image, out_alpha [H,W]. */
TN_API int tn_splat_raster(const TnSplatCamera* camera, int64_t num_gaussians, void* workspace, int64_t max_intersections, const float* background4,
                    int32_t antialiased, float* out_rgbt, float* out_depth, float* out_alpha, tn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* THERMAL_NERF_HIP_H */
