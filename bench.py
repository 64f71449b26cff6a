#!/usr/bin/env python3
"""bench.py -- train rays/s of thermal-nerfacto (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --mode separate --rays 8192          # BASELINE configs[2]
    python bench.py --path model-api                     # the reference Trainer's call sequence instead of the fused step

A "step" is one Trainer.train_iteration of the reference (engine/trainer.py:455-499) on a synthetic batch: pixel sampling + ray generation
for 4096 rays per GPU (configs[1]: density_mode=shared, 256/96 proposal + 48 field samples, 2^19/2^17 tables), forward, every loss,
backward, gradient all-reduce (N>1) and Adam over all parameter groups.  Inputs (cameras, images) are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# (no GPU_MAX_HW_QUEUES here: the schedules fit the runtime's default four hardware queues -- nerfstudio-thermal_amd/__init__.py; the
# environment's own setting, if any, is reported in the JSON as `hw_queues_env`)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

RAYS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md, "L2 (per XCD)": ~34.5 TB/s aggregate over the 8 XCDs
PMC_JSON = os.path.join(ROOT, "profiles", "r06_pmc.json")


def source_hash() -> str:
    """Hash of the kernel sources: PMC figures (profiles/*_pmc.json) are only quoted while they were measured on these very kernels."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "nerfstudio-thermal_amd", "csrc", "*.h*"))):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    return h.hexdigest()[:16]


def build_engine(device, mode="shared", seed=0, nerf_samples=48):
    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.arena import ParamArena
    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
    from nerfstudio_thermal_amd.engine import RenderEngine

    cfg = ThermalNerfactoModelConfig(density_mode=mode, num_nerf_samples_per_ray=nerf_samples)
    arena = ParamArena(cfg, 8, device)
    shapes = {n: s for n, (_, s) in arena.layout.items()}
    arena.load({k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, seed=seed).items()})
    eng = RenderEngine(cfg, arena, 8, [0, 0, 0, 0, 1, 1, 1, 1])
    return cfg, arena, eng


def build_model(device, mode="shared", seed=0, nerf_samples=48):
    """The drop-in object: ThermalNerfactoModel behind the reference's Model API, same synthetic weights as build_engine."""
    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
    from nerfstudio_thermal_amd.model import SceneBox

    cfg = ThermalNerfactoModelConfig(density_mode=mode, num_nerf_samples_per_ray=nerf_samples)
    model = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=8, metadata={"is_thermal": [0, 0, 0, 0, 1, 1, 1, 1]},
                      device=device)
    shapes = {n: s for n, (_, s) in model.arena.layout.items()}
    model.arena.load({k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, seed=seed).items()})
    model.train()
    return cfg, model.arena, model


def make_batch(device, num_rays, seed):
    """Pre-staged pixel batch (kernel_roofline and the parity-style fixed batch)."""
    from nerfstudio_thermal_amd import synth

    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, num_rays, seed=seed)
    img, is_th = synth.synth_gt(idx, cams, seed=seed)
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    cam_t = {k: t(cams[k]) for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    return cam_t, t(idx), t(img), t(is_th)


def make_image_cache(device):
    """The 8 synthetic training images (4 RGB 640x480, 4 thermal 160x120) resident in HBM, in dataset order."""
    from nerfstudio_thermal_amd import ops, synth

    cams = synth.synth_cameras()
    imgs = [torch.from_numpy(im) for im in synth.synth_images(cams)]
    n = len(imgs)
    return ops.ImageCache.build(imgs, torch.from_numpy(cams["is_thermal"].astype(np.float32)), torch.arange(n), device)


def _datamanager(owner, cam_t, cache, num_rays):
    dm = getattr(owner, "_bench_dm", None)
    if dm is None or dm.num_rays != num_rays or dm.cache is not cache:
        from nerfstudio_thermal_amd.data import DeviceDataManager

        dm = owner._bench_dm = DeviceDataManager(cache, cam_t, num_rays, 2)
    return dm


def one_step(eng, cam_t, cache, num_rays, step, hook, scaler=None):
    """Fused step.  datamanager.next_train (data/datamanagers/base_datamanager.py:538-547): this step's 2x2 pixel patches over the jagged image
    list, their ground truth and the rays, all on the device (nerfstudio_thermal_amd/data.py); then RenderEngine.train_step."""
    o, d, cam, img, is_th = _datamanager(eng, cam_t, cache, num_rays).next_train(step)
    return eng.train_step(o, d, cam, img, is_th, step, grad_hook=hook, grad_scaler=scaler)


def one_step_api(model, optimizers, cam_t, cache, num_rays, step, grad_scaler=None, call=None):
    """The reference Trainer's own sequence (engine/trainer.py:455-499) on the drop-in objects, statement for statement: callbacks,
    zero_grad_some, torch.autocast around model(ray_bundle) / get_metrics_dict / get_loss_dict (mixed_precision=True in thermal-nerfacto's method
    config), grad_scaler.scale(loss).backward(), optimizer_scaler_step_some, grad_scaler.update(), scheduler steps only when the scale did not
    drop.  `call`: the module to call for the forward (the DistributedDataParallel wrapper in multi-GPU runs)."""
    import functools

    from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc
    from nerfstudio_thermal_amd.rays import RayBundle

    o, d, cam, img, is_th = _datamanager(model, cam_t, cache, num_rays).next_train(step)
    cbs = model.__dict__.setdefault("_bench_cbs", model.get_training_callbacks())
    for cb in cbs:
        cb.run_callback_at_location(step, Loc.BEFORE_TRAIN_ITERATION)
    groups = model.__dict__.setdefault("_bench_groups", list(optimizers.optimizers.keys()))
    optimizers.zero_grad_some(groups)
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=cam[:, None])
    batch = {"image": img, "is_thermal": is_th}
    with torch.autocast(device_type="cuda", enabled=grad_scaler is not None and grad_scaler.is_enabled()):
        out = (call or model)(rb)
        metrics = model.get_metrics_dict(out, batch)
        losses = model.get_loss_dict(out, batch, metrics)
        loss = functools.reduce(torch.add, losses.values())
    if grad_scaler is None:
        loss.backward()
        optimizers.optimizer_step_all(step)
        optimizers.scheduler_step_all(step)
    else:
        grad_scaler.scale(loss).backward()
        optimizers.optimizer_scaler_step_some(grad_scaler, groups)
        scale = grad_scaler.get_scale()
        grad_scaler.update()
        if scale <= grad_scaler.get_scale():
            optimizers.scheduler_step_all(step)
    for cb in cbs:
        cb.run_callback_at_location(step, Loc.AFTER_TRAIN_ITERATION)
    return losses


class _TrainerShim:
    """What nerfstudio's Trainer holds, as far as trainer.FusedTrainerMixin reads it (engine/trainer.py:85-140): nerfstudio itself is not
    importable on the GPU box.  `step()` is the body of the Trainer's loop: callbacks, train_iteration, callbacks (engine/trainer.py:258-276)."""

    def __init__(self, model, optimizers, datamanager_next):
        import types

        self.pipeline = types.SimpleNamespace(model=model, datamanager=types.SimpleNamespace(next_train=datamanager_next))
        self.optimizers = optimizers
        self.mixed_precision = True
        self.grad_scaler = torch.amp.GradScaler("cuda")
        self.gradient_accumulation_steps = {g: 1 for g in optimizers.optimizers}
        self.config = types.SimpleNamespace(log_gradients=False)
        self.callbacks = model.get_training_callbacks()

    def train_iteration(self, step):
        raise RuntimeError("the fused trainer fell through to the reference iteration")

    def step(self, step):
        from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc

        for cb in self.callbacks:
            cb.run_callback_at_location(step, Loc.BEFORE_TRAIN_ITERATION)
        out = self.train_iteration(step)
        for cb in self.callbacks:
            cb.run_callback_at_location(step, Loc.AFTER_TRAIN_ITERATION)
        return out


def make_fused_trainer(model, cam_t, cache, rays):
    """The reference Trainer's loop body on trainer.FusedTrainerMixin with the DataManager adapter (what `ns-train thermal-nerfacto-hip` runs:
    plugin.py installs HipTrainer + the device datamanager): `pipeline.datamanager.next_train(step)` is datamanager.TrainRaySource.next -- one
    launch -> the reference's (RayBundle, batch) pair with every field (pixel_area, directions_norm, indices)."""
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.datamanager import TrainRaySource
    from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers
    from nerfstudio_thermal_amd.trainer import FusedTrainerMixin

    cams = synth.synth_cameras()
    src = TrainRaySource([torch.from_numpy(im) for im in synth.synth_images(cams)], cams["is_thermal"].astype(np.float32), cam_t, rays, 2, model.device,
                         shuffle=False)
    cls = type("HipTrainer", (FusedTrainerMixin, _TrainerShim), {})
    return cls(model, Optimizers(model.get_param_groups(), optimizer_cls=HipFusedAdam), src.next)


def time_ms(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# bench row -> kernels of profiles/r06_pmc.json ("<kernel> <grid X>x<grid Y>"), N = 4096 shared only.
# The DOMINANT row (the top-level `roofline` object) is derived from the measured stand-alone times x launches per step (rows of kernel_roofline:
# proposal-grid rows count only on update steps), not named here.
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak of MI355X (256 CUs x 4 x 64 FLOP/cycle/SIMD... MI355X_MICROARCH.md); scripts/microbench/mfma_rate.hip sustains 155
# kernel name + points of the launch; the key in the PMC file also carries the grid's y extent (level groups: "k_seg_bin<false> 196608x4"),
# which changes with the block size of the bin pass -- matched by prefix
PMC_KEYS = {
    "scatter(main grid)": ["k_seg_bin<false> 196608x", "k_seg_fold 196608x"],
    "scatter(prop0 grid)": ["k_seg_bin<true> 1048576x", "k_seg_fold 1048576x"],
    "scatter(prop1 grid)": ["k_seg_bin<true> 393216x", "k_seg_fold 393216x"],
    "k_prop_fwd(level0)": ["k_prop_fwd 1048576x"],
    "k_prop_fwd(level1)": ["k_prop_fwd 393216x"],
}


def pmc_lookup(pmc, prefixes):
    """the PMC entries of a bench row (one per kernel of the entry point), or None when one is missing"""
    out = []
    for pre in prefixes:
        hit = [v for k, v in pmc.items() if k.startswith(pre)]
        if len(hit) != 1:
            return None
        out.append(hit[0])
    return out


def load_pmc():
    """PMC figures measured by scripts/pmc_passes.sh, or (None, why) when they were taken on other kernel sources than the ones running."""
    if not os.path.exists(PMC_JSON):
        return None, "no profiles/r06_pmc.json"
    with open(PMC_JSON) as f:
        j = json.load(f)
    if j.get("source_hash") != source_hash():
        return None, f"profiles/r06_pmc.json was measured on kernel sources {j.get('source_hash')}, running {source_hash()}: re-run scripts/pmc_passes.sh"
    return j["kernels"], None


def kernel_roofline(eng, cam_t, idx, image=None, is_thermal=None):
    """Live HIP-event timing (torch.cuda.Event on torch's current stream = the stream every kernel of this library is launched on) of the
    hash-grid gather / scatter entry points, each launched alone as a single C-ABI call.  Algorithmic bytes: gather =
    points x levels x 8 corners x 8 B; scatter-add = read-modify-write = 2 x that.
    image / is_thermal (the batch's ground truth): the main grid's scatter -- the dominant launch pair -- is then timed on the step's OWN d enc
    (real losses -> renderer backward -> k_field_bwd_fused leave it in the field's workspace; the scatter phase of tn_field_bwd_phase reads it there):
    the fold skips all-zero pairs, and the zero pattern of real gradients is not that of random ones."""
    from nerfstudio_thermal_amd import ops

    N = idx.shape[0]
    o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
    cam = idx[:, 0].contiguous()
    out, br = eng.get_outputs(o, d, cam, True)
    b = br[""]
    lv = b.levels
    rows = []
    for i in range(2):
        S = lv[i].S
        ms = time_ms(lambda i=i: ops.prop_density_fwd(eng.props[i], b.origins, b.directions, lv[i].e_bins))
        rows.append((f"k_prop_fwd(level{i})", ms, N * S * 5 * 8 * 8))
    d_o, d_d = torch.zeros_like(o), torch.zeros_like(d)
    grids = [("prop0 grid", eng.props[0], lv[0]), ("prop1 grid", eng.props[1], lv[1]), ("main grid", eng.field, lv[2])]
    for name, net, L in grids:
        # level-major [L][P] float2, as the backward kernels of all three grids hand their d enc over
        g_enc = (torch.randn((N * L.S, net.num_levels, 2), device=o.device) * 1e-3).permute(1, 0, 2).contiguous()
        # as the step calls it: the proposal grids' scatter also yields d position; the main field's does not (k_field_dpos does, beside it);
        # shared mode: the step's optimiser launch leaves the gradients zero and every table sees one scatter per iteration, so the fold stores
        # instead of adding (TnGrid.table_grad_is_zero) -- timed that way here
        dpos = (d_o, d_d) if net.num_levels == 5 else (None, None)
        if net.num_levels != 5 and image is not None and not eng.separate:  # (shared density: one RGBT composite, the step bench.py's `value` times)
            c = eng.cfg
            d_comp, dw2 = torch.zeros((N, 4), device=o.device), torch.zeros((N, L.S), device=o.device)
            lines = torch.zeros((ops.LOSS_LINES, 16), device=o.device)
            pixel = (b.comp[:, :3], b.comp[:, 3:], image, is_thermal, c.thermal_loss_mult, c.tv_pixel_loss_mult, c.cross_channel_loss_mult, d_comp[:, :3], d_comp[:, 3:])
            ops.train_losses(L.s_bins, L.weights, [(lv[i].s_bins, lv[i].weights, None) for i in range(2)], c.distortion_loss_mult, c.interlevel_loss_mult, dw2, lines,
                             pixel=pixel)
            d_rgb, d_dens = ops.render_bwd(L.e_bins, L.density, b.rgb_samples, L.weights, d_comp, dw2)
            ph = ops._lib
            ops.field_bwd_phase(net, b.origins, b.directions, cam, L.e_bins, d_dens, d_rgb, None, None, ph.TN_BWD_MLP | ph.TN_BWD_JOIN)  # d enc -> workspace
            eng._set_grad_zero(not eng.separate)  # (shared mode: the fold stores, as in the step)
            try:
                ms = time_ms(lambda: ops.field_bwd_phase(net, b.origins, b.directions, cam, L.e_bins, d_dens, d_rgb, None, None, ph.TN_BWD_SCATTER | ph.TN_BWD_JOIN,
                                                         0, net.num_levels))
            finally:
                eng._set_grad_zero(False)
        else:
            ms = time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions,
                                                  L.e_bins, g_enc, *dpos, grad_is_zero=not eng.separate))
        rows.append((f"scatter({name})", ms, 2 * N * L.S * net.num_levels * 8 * 8))
    # ---- the MFMA-bound launches of the main field, stand-alone: forward (prep + XCD-affine gather + chain) and the backward's MLP phase
    # (k_field_bwd_fused: chain + every weight gradient).  FLOPs are ALGORITHMIC: per sample the base MLP 2 (32 64 + 64 16) = 6 144
    # and the head 2 (63 64 + 64 64 + 64 4) = 16 768 forward; the backward (d input + d weight of every layer) = twice that.
    S2 = lv[2].S
    P2 = N * S2
    fwd_flop = P2 * (6144 + 16768)
    ms_f = time_ms(lambda: ops.field_fwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, training=True))
    gd, gc = torch.rand_like(lv[2].density), torch.rand_like(b.rgb_samples)
    ms_b = time_ms(lambda: ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, None, None, ops._lib.TN_BWD_MLP | ops._lib.TN_BWD_JOIN))
    mfma = [("tn_field_fwd (k_field_prep + k_field_encode_xcd + k_field_mlp_fwd)", ms_f, fwd_flop, 160 * 4096),
            ("k_field_bwd_fused (MLP phase of tn_field_bwd)", ms_b, 2 * fwd_flop, 344 * 4096)]
    eng.arena.zero_grad()
    tiles = (P2 + 31) // 32
    return rows, [{"kernel": n, "ms": m, "flop_per_launch": fl, "achieved_tflops": fl / (m * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                   "frac": fl / (m * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "issued_mfma_flop_per_launch": tiles * per_tile,
                   "note": "flop_per_launch = algorithmic (SURVEY.md 8d); issued = MFMA instructions x their FLOPs (the embedding's share was removed by algebra); "
                           "the forward launch group also holds the HBM/L2-bound gather (about half of its time)"}
                  for n, m, fl, per_tile in mfma]


def step_algorithmic_bytes(arena, mode, rays, update_frac, nerf_samples=48):
    """(N x bytes_ray + bytes_step) of one train step, no cache credit.  Per ray: forward gathers 161 792 B (separate: 421 888 B
    incl. the two cross-evaluated densities); backward scatter-add 98 304 B per main-grid backward (shared 1, separate 4: two branches + two
    cross terms) and 225 280 B per proposal-network backward (shared: on update steps; separate: the thermal sampler updates every step).
    Per step: Adam 28 B per optimised parameter (proposal groups only when they are stepped)."""
    field = nerf_samples * 16 * 8 * 8  # one main-grid gather per ray (48 samples: 49 152 B)
    prop = (256 + 96) * 5 * 8 * 8  # both proposal levels
    fwd = prop + field if mode == "shared" else 2 * prop + 4 * field  # separate: two branches + two cross-evaluated densities
    main_bwd = 2 * field * (1 if mode == "shared" else 4)
    prop_bwd = 225280 * (update_frac + (1.0 if mode == "separate" else 0.0))
    n_prop = sum(int(np.prod(arena.layout[k][1])) for k in arena.group_keys["proposal_networks"])
    n_all = arena.num_optimised()
    adam = 28.0 * (n_all - (1.0 - update_frac) * n_prop)
    return rays * (fwd + main_bwd + prop_bwd) + adam


def cpu_baseline(num_rays, steps, threads, mode="shared"):
    """Oracle (pure-PyTorch restatement of the reference torch path, pinned to reference goldens) timed on the host cores:
    forward + losses + backward + Adam, same workload definition, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import thermal_nerfacto_oracle as orc
    from nerfstudio_thermal_amd import synth

    torch.set_num_threads(threads)
    cfg = orc.OracleConfig(density_mode=mode)
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.synth_params(orc.param_shapes(cfg), seed=0).items()}
    cams = synth.synth_cameras()
    idx = torch.from_numpy(synth.synth_ray_indices(cams, num_rays, seed=42))
    img, is_th = (torch.from_numpy(a) for a in synth.synth_gt(idx.numpy(), cams, seed=42))
    tc = {k: torch.from_numpy(cams[k]) for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    groups = orc.optimizer_groups(cfg)
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
    times = []
    for step in range(steps + 1):
        t0 = time.perf_counter()
        o, d, _, _ = orc.generate_rays(idx, tc["c2w"], tc["fx"], tc["fy"], tc["cx"], tc["cy"], tc["distortion"])
        jit = [torch.rand(num_rays, 1) for _ in range(3)]
        jit_t = [torch.rand(num_rays, 1) for _ in range(3)]
        out = orc.get_outputs(params, cfg, o, d, idx[:, 0], training=True, anneal=1.0, jitters=jit, jitters_thermal=jit_t)
        losses = orc.loss_dict(params, cfg, out, img, is_th, training=True)
        sum(losses.values()).backward()
        with torch.no_grad():
            for _, (keys, lr) in groups.items():
                for k in keys:
                    p = params[k]
                    if p.grad is not None:
                        orc.adam_step(p, p.grad, state[k][0], state[k][1], step + 1, lr)
                        p.grad = None
        if step > 0:
            times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return num_rays / t, t


def bench_splat(args):
    """BASELINE configs[4]: thermal-splatfacto forward render at 1080p (N4, parity unpinned).  One step = one frame: project + SH colours,
    tile binning (depth sort, pair emission, tile sort), raster of RGB + thermal + depth + accumulation; Gaussians resident in HBM.
    Replicas only across GPUs (every rank renders its own frames: a frame does not shard in the reference either)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerfstudio_thermal_amd  # noqa: F401
    import splat_oracle as so  # the CPU baseline leg only (checker, never the thing measured)
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.parallel import init_distributed
    from nerfstudio_thermal_amd.splat import PinholeCamera, ThermalSplatfactoModel, ThermalSplatfactoModelConfig

    rank, local, world = init_distributed()
    torch.cuda.set_device(local)
    N, W, H = args.gaussians, 1920, 1080
    p = synth.synth_gaussians(N, seed=11 + rank, extent=1.5, scale_range=(-5.5, -3.5))
    m = ThermalSplatfactoModel(ThermalSplatfactoModelConfig(), num_points=4, device=f"cuda:{local}")
    m.load_gaussians(p)
    m.step = 10**6  # all SH degrees active
    cam = PinholeCamera(synth.look_at_camera((3.2, 0.5, 0.8)), 1400.0, 1400.0, 960.0, 540.0, W, H)
    for _ in range(args.warmup):
        m.get_outputs(cam)
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = m.get_outputs(cam)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=f"cuda:{local}", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    # raster alone, HIP events on torch's current stream (the stream the library launches on)
    import ctypes as C

    from nerfstudio_thermal_amd import _lib
    from nerfstudio_thermal_amd.ops import _stream
    from nerfstudio_thermal_amd.splat import camera_struct

    cs = camera_struct(cam)
    rgbt, dep, alp = (torch.empty((H, W, c), device=f"cuda:{local}") for c in (4, 1, 1))
    bg4 = (C.c_float * 4)(0, 0, 0, 0)
    lib = _lib.load()

    def raster():
        # (depth_max is only reset by tn_splat_bin: re-running the raster alone re-derives the same images)
        _lib.check(lib.tn_splat_raster(C.byref(cs), N, C.c_void_p(m._ws.data_ptr()), m._cap, bg4, 0, C.c_void_p(rgbt.data_ptr()), C.c_void_p(dep.data_ptr()),
                                       C.c_void_p(alp.data_ptr()), _stream()), "tn_splat_raster")

    t_r = time_ms(raster, iters=10, warmup=2)
    line = {
        "metric": "thermal-splatfacto forward render, frames/s at 1080p", "value": world * 1e3 / ms, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"thermal-splatfacto (RGB+T Gaussians) forward render, 1920x1080, 16x16 tiles, {N} synthetic Gaussians, degree-3 SH, classic mode",
                   "visible": int((m.last_projection["radii"] > 0).sum()), "tile_pairs": m.last_num_intersections, "parallelism": f"replicas x{world}",
                   "parity": "unpinned (gsplat outside the reference tree; oracle = published algorithm)"},
        "roofline": {"bound": "valu", "kernel": "k_splat_raster", "avg_launch_ms": t_r, "achieved": None, "peak": None, "unit": "VALU issue", "frac": None,
                     "traffic": None, "note": "VALU-issue bound (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration) in profiles/r02_splat_pmc.md); neither HBM- nor MFMA-bound"},
        "Mpix_per_s": W * H / ms / 1e3, "mean_accumulation": float(out["accumulation"].mean()),
    }
    if rank == 0 and not args.no_cpu_baseline:
        # the oracle (port) on a BOUNDED sample: 1/64 of the frame (240x135, intrinsics scaled) with 1/64 of the Gaussians, host cores
        n_s, Ws, Hs = max(N // 64, 1000), W // 8, H // 8
        ps = {k: v[:n_s] for k, v in p.items()}
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        t0 = time.perf_counter()
        so.render(ps, synth.look_at_camera((3.2, 0.5, 0.8)), 1400.0 / 8, 1400.0 / 8, 960.0 / 8, 540.0 / 8, Ws, Hs)
        tc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": 1.0 / (tc * 64.0), "unit": "frames/s (extrapolated x64 from the sample)", "cores": torch.get_num_threads(), "kind": "port",
                                "sample": f"{Ws}x{Hs} pixels, {n_s} Gaussians by oracle/splat_oracle.py in {tc:.1f} s"}
    if rank == 0:
        print(json.dumps(line))


def parity_vs_reference(eng, device):
    """The "render PSNR vs ref" half of BASELINE.json's metric, outside any timed region: the eval-mode render of the 256 rays of
    tests/golden/model_shared_default256.npz -- outputs of the REFERENCE's own ThermalNerfactoModel(implementation="torch") at the default table
    sizes on the same synthetic weights (oracle/make_golden.py --default --rays 256) -- against this engine's render of the same rays."""
    path = os.path.join(ROOT, "tests", "golden", "model_shared_default256.npz")
    g = np.load(path)
    t = lambda k, dt=None: torch.from_numpy(g[k] if dt is None else g[k].astype(dt)).to(device).contiguous()  # noqa: E731
    o, d, cam = t("rays/origins"), t("rays/directions"), t("rays/camera_indices")[:, 0].contiguous()
    out, _ = eng.get_outputs(o, d, cam, training=False)
    torch.cuda.synchronize()

    def cmp(key):
        a, b = out[key].detach().cpu().double(), torch.from_numpy(g[f"eval/{key}"]).double()
        err = (a - b).abs()
        mse = float((err ** 2).mean())
        return {"max_abs": float(err.max()), "psnr_db": (10.0 * float(np.log10(1.0 / mse)) if mse > 0 else float("inf"))}

    rgb, th, den = cmp("rgb"), cmp("rgb_thermal"), cmp("density")
    derr = (out["density"].detach().cpu().double() - torch.from_numpy(g["eval/density"]).double()).abs()
    res = {"against": "tests/golden/model_shared_default256.npz = outputs of the reference itself (torch path), 256 rays, 16x2^19 + 2x(5x2^17) tables, eval render",
           "psnr_rgb_db": rgb["psnr_db"], "psnr_thermal_db": th["psnr_db"], "max_abs_rgb": rgb["max_abs"], "max_abs_thermal": th["max_abs"],
           "max_abs_density": den["max_abs"], "density_frac_above_1e-4": float((derr > 1e-4).double().mean()),
           "tolerances": {"rgb_thermal_abs": 1e-3, "density_abs": 1e-4, "note": "density after the WHOLE sampling chain is conditioned by the "
                          "reference's own 1-ulp response (1.3e-4, tests/test_conditioning_cpu.py); strict 1e-4 holds on identical samples (tests/test_model_gpu.py)"}}
    # ---- conditioning of the chain, tracked per round: the error as a multiple of the reference arithmetic's own response to ONE ulp of its field
    # sample bins (tests/golden/conditioning.json: data written by oracle/make_conditioning.py; the test bound is K_CHAIN = 6), its share above 1e-4
    # against the reference's share under +-4 ulps, and how many of the HIP sampler's resampled bins sit more than 64 ulps from the reference's
    # (train forward with the golden's jitter: a bin that fell into the neighbouring CDF interval)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "conditioning.json")) as f:
            cond = json.load(f)["shared/default256"]
        res["density_err_over_1ulp_response"] = den["max_abs"] / cond["ulp1_max"]
        res["density_share_above_1e-4_over_reference_share_at_4ulp"] = res["density_frac_above_1e-4"] / cond["ulp4_frac_above_1e-4"]
        res["conditioning"] = {"reference_1ulp_max": cond["ulp1_max"], "reference_4ulp_share_above_1e-4": cond["ulp4_frac_above_1e-4"], "test_bound_K_CHAIN": 6.0}
        from nerfstudio_thermal_amd import synth

        n = int(g["num_rays"])
        jit = [torch.from_numpy(j).to(device).reshape(-1).contiguous() for j in synth.synth_jitters(n)]
        saved = eng.anneal
        eng.set_anneal_for_step(500)
        assert abs(eng.anneal - float(g["train/anneal"])) < 1e-12
        _, br = eng.get_outputs(o, d, cam, True, jit, None)
        eng.anneal = saved
        shares = {}
        for i, lv in enumerate(br[""].levels):
            a = lv.e_bins.detach().cpu().contiguous().view(torch.int32).to(torch.int64)
            b = torch.from_numpy(np.ascontiguousarray(g[f"train/ebins_{i}"])).view(torch.int32).to(torch.int64)
            dist = (a - b).abs()  # positive floats: the int32 views are monotone in the value
            shares[f"level{i}"] = {"identical": float((dist == 0).double().mean()), "within_4ulp": float((dist <= 4).double().mean()),
                                   "beyond_64ulp": float((dist > 64).double().mean())}
        res["sampler_bins_vs_reference_ulp"] = shares
        res["bins_beyond_64ulp_share"] = shares["level2"]["beyond_64ulp"]
    except Exception as e:  # noqa: BLE001  (diagnostics never cost the parity block)
        res["conditioning_error"] = repr(e)
    return res


def extra_leg(device, mode, rays, nerf_samples, path, steps, warmup):
    """One of the other BASELINE configs as extra keys of the default line (never part of `value`): a fresh engine / model, `warmup` + `steps`
    steps timed with a host clock around a synchronise, and the main grid's scatter entry point stand-alone on that workload."""
    from nerfstudio_thermal_amd.optim import DeviceGradScaler

    api = path == "model-api"
    cam_t, idx, _, _ = make_batch(device, rays, seed=42)
    cache = make_image_cache(device)
    if path == "fused-trainer":
        cfg, arena, model = build_model(device, mode=mode, nerf_samples=nerf_samples)
        eng = model.engine
        trainer = make_fused_trainer(model, cam_t, cache, rays)
        run = trainer.step
    elif api:
        from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers

        cfg, arena, model = build_model(device, mode=mode, nerf_samples=nerf_samples)
        eng = model.engine
        optimizers = Optimizers(model.get_param_groups(), optimizer_cls=HipFusedAdam)
        scaler = torch.amp.GradScaler("cuda")
        run = lambda st: one_step_api(model, optimizers, cam_t, cache, rays, st, scaler)  # noqa: E731
    else:
        cfg, arena, eng = build_engine(device, mode=mode, nerf_samples=nerf_samples)
        scaler = DeviceGradScaler(device, num_groups=len(arena.optimised_groups))
        run = lambda st: one_step(eng, cam_t, cache, rays, st, None, scaler)  # noqa: E731
    step = 0
    for _ in range(warmup):
        run(step)
        step += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    upd = 0
    for _ in range(steps):
        run(step)
        upd += int(eng.steps_since_update == 1)
        step += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    opt_in = {}
    if api:
        # the same sequence with the opt-in host setting of nerfstudio_thermal_amd.configure_host(): autograd's nodes on the calling thread
        import nerfstudio_thermal_amd as pkg

        pkg.configure_host(single_thread_backward=True)
        try:
            for _ in range(warmup):
                run(step)
                step += 1
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                run(step)
                step += 1
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
        finally:
            pkg.configure_host(single_thread_backward=False)
        opt_in = {"single_thread_backward": {"ms_per_step": dt1 / steps * 1e3, "rays_per_s": rays * steps / dt1,
                                             "note": "torch.autograd.set_multithreading_enabled(False) (nerfstudio_thermal_amd.configure_host, opt-in); "
                                                     "ms_per_step above is with torch's defaults"}}
    # (the main grid's scatter entry point on THIS workload: the headline line's dominant kernel, for comparison across the legs)
    name, ms, nbytes = next(r for r in kernel_roofline(eng, cam_t, idx)[0] if r[0] == "scatter(main grid)")
    how = {"model-api": " (autocast + torch.amp.GradScaler + HipFusedAdam: the reference Trainer's sequence, engine/trainer.py:455-499)",
           "fused-trainer": " (the reference Trainer's loop body -- callbacks, train_iteration, callbacks -- with train_iteration on the fused step: "
                            "trainer.FusedTrainerMixin, what the method plugin's TrainerConfig._target runs)"}.get(path, " (device-side GradScaler)")
    return {**opt_in, "workload": f"density_mode={mode}, {rays} rays, {nerf_samples} field samples, path {path}" + how,
            "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "rays_per_s": rays * steps / dt, "proposal_update_fraction": upd / steps,
            "dominant_kernel": name, "dominant_kernel_ms": ms, "dominant_kernel_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def head_bf16x3_leg(device, steps=40, warmup=10):
    """OPT-IN EXPERIMENT, never part of `value` and never the default: TN_HEAD_BF16X3=1 runs the colour head's two 64-wide layers
    (fields/thermal_nerfacto_field.py:91-99, field_components/mlp.py:159-178) -- forward, backward chain and weight gradients -- on split-bf16
    matrix instructions (x = hi + lo, three v_mfma_f32_32x32x16_bf16 per product, fp32 accumulators); the density path and every other kernel stay
    fp32.  The leg times the headline's fused step with and without the switch (fresh engines, same seeds) and carries its own parity block:
    the eval render of the reference golden under the switch, and the gradients of one training step on one batch against the fp32 path's."""
    from nerfstudio_thermal_amd.optim import DeviceGradScaler

    prev = os.environ.get("TN_HEAD_BF16X3")
    res = {"dtype": "bf16x3 (hi + lo split bf16, fp32 accumulate) in the colour head's 64-wide layers; f32 everywhere else",
           "switch": "TN_HEAD_BF16X3=1 (read per call by tn_field_fwd / tn_field_bwd*)"}
    try:
        cam_t, idx, _, _ = make_batch(device, RAYS_PER_GPU, seed=42)
        cache = make_image_cache(device)
        grads, timing = {}, {}
        for flag in ("0", "1"):
            os.environ["TN_HEAD_BF16X3"] = flag
            torch.manual_seed(20261)  # (the pixel sampler's and the ray samplers' uniforms: the same batch and jitter for both paths)
            cfg, arena, eng = build_engine(device)
            scaler = DeviceGradScaler(device, num_groups=len(arena.optimised_groups))
            if flag == "1":
                par = parity_vs_reference(eng, device)
                res["parity"] = {k: par[k] for k in ("against", "psnr_rgb_db", "psnr_thermal_db", "max_abs_rgb", "max_abs_thermal", "max_abs_density")}
            # one training step on one batch, gradients read behind the backward (a plain hook: the five-call sequence, same kernels)
            keep = {}

            def hook(a, keep=keep):
                for name in a.names():
                    keep[name] = a.grad_view(name).detach().clone()

            if hasattr(eng, "_bench_dm"):
                del eng._bench_dm
            one_step(eng, cam_t, cache, RAYS_PER_GPU, 0, hook, scaler)
            torch.cuda.synchronize()
            grads[flag] = keep
            del eng, arena
            torch.manual_seed(20262)
            cfg, arena, eng = build_engine(device)
            scaler = DeviceGradScaler(device, num_groups=len(arena.optimised_groups))
            step = 0
            for _ in range(warmup):
                one_step(eng, cam_t, cache, RAYS_PER_GPU, step, None, scaler)
                step += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            upd = 0
            for _ in range(steps):
                one_step(eng, cam_t, cache, RAYS_PER_GPU, step, None, scaler)
                upd += int(eng.steps_since_update == 1)
                step += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            timing[flag] = {"ms_per_step": dt / steps * 1e3, "rays_per_s": RAYS_PER_GPU * steps / dt, "proposal_update_fraction": upd / steps}
            del eng, arena
            torch.cuda.empty_cache()
        res.update({"steps": steps, "warmup": warmup, "ms_per_step": timing["1"]["ms_per_step"], "rays_per_s": timing["1"]["rays_per_s"],
                    "proposal_update_fraction": timing["1"]["proposal_update_fraction"], "f32_same_run": timing["0"],
                    "speedup_over_f32_same_run": timing["0"]["ms_per_step"] / timing["1"]["ms_per_step"]})
        worst = {}
        for name, ref in grads["0"].items():
            scale = float(ref.abs().max())
            if scale == 0.0:
                continue
            worst[name] = float((grads["1"][name] - ref).abs().max()) / scale
        head = {k: v for k, v in worst.items() if ".mlp_head." in k or "embedding_appearance" in k}
        res["gradients_vs_f32_path"] = {
            "what": "max |g_bf16x3 - g_f32| / max |g_f32| per parameter, one training step of 4096 rays on identical weights and batch",
            "head_max": max(head.values()) if head else None, "all_max": max(worst.values()), "worst_parameter": max(worst, key=worst.get),
            "note": "the backward's own arithmetic agrees to ~5e-6 on identical saved activations (tests/test_head_bf16x3_gpu.py); what is left here is "
                    "the forward's ~1e-5 moving a few pre-activations across zero -- a ReLU mask flip changes a gradient by one sample's contribution"}
    finally:
        if prev is None:
            os.environ.pop("TN_HEAD_BF16X3", None)
        else:
            os.environ["TN_HEAD_BF16X3"] = prev
    return res


def eval_render_leg(device, reps=3, profile_path=None):
    """TEST_RAYS_PER_SEC of the reference (utils/writer.py:55-56, engine/trainer.py:519-527): one full RGB image (640x480) + one full thermal image
    (160x120) of the synthetic scene's cameras through `Model.get_outputs_for_camera` (models/base_model.py:165-205: chunks of
    eval_num_rays_per_chunk = 32768 rays) -- and the same rays through `tn_render_rays_eval` directly, chunk by chunk.  Algorithmic bytes per ray =
    the forward gather term of SURVEY 8d (161 792 B).  Never part of `value`."""
    import types

    from nerfstudio_thermal_amd import ops, synth

    cfg, arena, model = build_model(device)
    model.config.eval_num_rays_per_chunk = 1 << 15  # method_configs["thermal-nerfacto"] (configs/method_configs.py:255-310), as plugin.py sets it
    model.eval()
    cams = synth.synth_cameras()

    def camera(i):
        return types.SimpleNamespace(camera_to_worlds=torch.from_numpy(cams["c2w"][i]), fx=float(cams["fx"][i]), fy=float(cams["fy"][i]), cx=float(cams["cx"][i]),
                                     cy=float(cams["cy"][i]), width=int(cams["width"][i]), height=int(cams["height"][i]),
                                     distortion_params=torch.from_numpy(cams["distortion"][i]), camera_index=i)

    pair = [camera(0), camera(4)]
    rays = sum(c.width * c.height for c in pair)

    def through_model():
        outs = [model.get_outputs_for_camera(c) for c in pair]
        return outs

    through_model()  # warm-up: workspaces, lin tables
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = through_model()
    torch.cuda.synchronize()
    dt_model = (time.perf_counter() - t0) / reps
    assert all(bool(torch.isfinite(o["rgb"]).all()) for o in outs)

    # the same rays through the C ABI's inference entry point, chunk by chunk (what get_outputs_for_camera_ray_bundle needs per chunk)
    eng = model.engine
    chunk = int(cfg.eval_num_rays_per_chunk)
    bundles = []
    for c in pair:
        H, W = c.height, c.width
        yy, xx = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
        idx = torch.stack([torch.zeros(H * W, dtype=torch.int64, device=device), yy.reshape(-1), xx.reshape(-1)], dim=1).contiguous()
        g = lambda v: torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1)  # noqa: E731
        o, d, _, _ = ops.raygen(idx, c.camera_to_worlds.to(device).reshape(1, 3, 4).contiguous(), g(c.fx), g(c.fy), g(c.cx), g(c.cy),
                                c.distortion_params.to(device).reshape(1, 6).contiguous())
        bundles.append((o, d, torch.full((H * W,), c.camera_index, dtype=torch.int64, device=device)))

    def direct():
        for o, d, cam in bundles:
            for i in range(0, o.shape[0], chunk):
                n = min(chunk, o.shape[0] - i)
                nears, fars = eng._nears_fars(n, False)
                ops.render_rays_eval(eng.props, eng.field, o[i:i + n], d[i:i + n], cam[i:i + n], nears, fars, eng.counts, eng.anneal)

    direct()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        direct()
    torch.cuda.synchronize()
    dt_direct = (time.perf_counter() - t0) / reps
    if profile_path:
        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CPU]) as prof:
            through_model()
            torch.cuda.synchronize()
        with open(profile_path, "w") as f:
            f.write(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40))
    bytes_ray = 161792.0
    return {"workload": "one 640x480 RGB + one 160x120 thermal image of the synthetic scene (326 400 rays), shared density, eval mode, chunks of %d rays" % chunk,
            "rays": rays, "reps": reps,
            "get_outputs_for_camera": {"ms": dt_model * 1e3, "rays_per_s": rays / dt_model, "frac_of_hbm_roofline": rays / dt_model * bytes_ray / 1e9 / HBM_PEAK_GBS},
            "tn_render_rays_eval": {"ms": dt_direct * 1e3, "rays_per_s": rays / dt_direct, "frac_of_hbm_roofline": rays / dt_direct * bytes_ray / 1e9 / HBM_PEAK_GBS},
            "algorithmic_bytes_per_ray": bytes_ray,
            "note": "the reference's TEST_RAYS_PER_SEC (utils/writer.py:55-56); gather term only, no cache credit: the proposal tables are L2-resident"}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks ourselves, as the reference's launcher does
    (scripts/train.py:138-151,204-209: mp.spawn of one process per device).  The parent has made NO HIP call at this point (torch is imported,
    nothing else): it starts `python -m torch.distributed.run` as a fresh child process, which starts one fresh process per GPU, and
    returns the child's exit code.  No process that has touched the GPU is ever replaced."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    env["TN_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(args):
    """--rendezvous-only: the launcher logic without the workload (CPU test of `--gpus N`): join the process group (gloo without a GPU), prove
    the rank count with a real all-reduce, print the same bookkeeping keys as the bench line."""
    import torch.distributed as dist

    from nerfstudio_thermal_amd.parallel import init_distributed

    rank, local, world = init_distributed()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    ranks = 1
    if world > 1:
        t = torch.ones(1)
        if dist.get_backend() == "nccl":
            t = t.cuda(local)
        dist.all_reduce(t)
        ranks = int(t.item())
        assert ranks == dist.get_world_size()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rccl_ranks": ranks, "backend": dist.get_backend() if world > 1 else None,
                          "spawned_by_bench": os.environ.get("TN_BENCH_SPAWNED") == "1"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="nerfacto", choices=["nerfacto", "splat"], help="nerfacto: the thermal-nerfacto train step (BASELINE metric); "
                    "splat: thermal-splatfacto forward render at 1080p (BASELINE configs[4], N4)")
    ap.add_argument("--gaussians", type=int, default=1_000_000, help="--workload splat: number of synthetic Gaussians")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="shared", choices=["shared", "separate"], help="density_mode (default = BASELINE configs[1]; separate = configs[2])")
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU per step (configs[2] uses 8192)")
    ap.add_argument("--nerf-samples", type=int, default=48, help="field samples per ray (num_nerf_samples_per_ray: 48 = the method's default; "
                    "96 = the variant SURVEY.md section 8 quotes the metric on)")
    ap.add_argument("--path", default="fused", choices=["fused", "model-api"], help="fused: RenderEngine.train_step (no autograd tape); model-api: the "
                    "reference Trainer's sequence forward -> get_metrics_dict -> get_loss_dict -> backward -> optimisers on ThermalNerfactoModel")
    ap.add_argument("--api-optimizer", default="hip", choices=["hip", "torch"], help="--path model-api: HipFusedAdam (one launch per group over the arena) or "
                    "torch.optim.Adam on the same parameters")
    ap.add_argument("--api-single-thread-backward", action="store_true", help="--path model-api: nerfstudio_thermal_amd.configure_host() -- autograd runs the "
                    "backward nodes on the calling thread (torch.autograd.set_multithreading_enabled(False)); an opt-in process setting, not torch's default")
    ap.add_argument("--no-grad-scaler", action="store_true", help="drop the GradScaler semantics of the reference Trainer (mixed_precision=True in thermal-nerfacto's "
                    "method config): fused path = no non-finite check / device-side skip (optim.DeviceGradScaler), model-api path = plain backward + step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs of the default run (configs[2] separate / 8192 rays, the 96-field-sample "
                    "variant, the drop-in path under the reference Trainer's AMP sequence) and the parity figures against the reference golden")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads for the baseline (0 = min(host cores, 16): more threads make the"
                    " many small ATen ops of this path slower; measured on the 256-thread GPU-box host at 1024 rays: 8 -> 2427, 16 -> 2286, 32 -> 1906, 64 -> 1025,"
                    " 256 -> 24 rays/s)")
    ap.add_argument("--ops", action="store_true", help="print the per-kernel timing table to stderr")
    ap.add_argument("--dp-chunks", type=int, default=-1, help="N>1: level ranges of the main table exchanged separately (-1 = the default "
                    "schedule, n = n equal ranges, 0 = one all-reduce after the backward)")
    ap.add_argument("--dp-adam-per-range", action="store_true", help="N>1 / --force-dp: one Adam launch per exchanged range instead of one after the exchange")
    ap.add_argument("--dp-shard-optimizer", action="store_true", help="N>1: reduce-scatter the big gradient slices, Adam on the owned 1/N piece, all-gather "
                    "the parameters (parallel.ShardedGradReducer; runs without the GradScaler semantics)")
    ap.add_argument("--dp-bf16", action="store_true", help="N>1: the big gradient slices travel as bfloat16 (half the bytes per xGMI link; changes numerics)")
    ap.add_argument("--force-dp", action="store_true", help="diagnostic: run the N>1 schedule (phased backward + overlapped RCCL all-reduce) on a "
                    "1-rank process group, to see what the schedule itself costs")
    ap.add_argument("--force-dp-sum", action="store_true", help="--force-dp: let the one-rank group exchange with SUM (RCCL launches nothing in place) instead of "
                    "AVG (a real RCCL kernel beside the folds, what N > 1 runs): the schedule's own cost without any collective kernel")
    ap.add_argument("--no-dp-guard", action="store_true", help="N>1 / --force-dp: skip parallel.ScheduleGuard (a few plain and a few overlapped steps are timed before "
                    "the warm-up, and a few of the simple schedule; the faster of the two real schedules runs -- switched in process)")
    ap.add_argument("--rendezvous-only", action="store_true", help="launcher test: spawn / join the ranks, all-reduce once, print the rank count, exit")
    ap.add_argument("--long-steps", type=int, default=200, help="extra steps after the timed region for the BASELINE.md 3.4 figure (median of >= 200 "
                    "per-step times, reported as extra keys; `value` always follows --steps).  0 disables")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # before any GPU call of this process
    if args.rendezvous_only:
        return rendezvous_only(args)
    if args.workload == "splat":
        return bench_splat(args)

    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import _lib
    from nerfstudio_thermal_amd.parallel import GradAllReducer, OverlappedGradReducer, broadcast_params, init_distributed, rank_seed

    _lib.load()  # fail loudly if the HIP library is missing
    rank, local, world = init_distributed()
    if world != args.gpus:  # never degrade silently to fewer ranks than asked for
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without a "
              "torchrun environment (bench.py then spawns the ranks itself)", file=sys.stderr)
        sys.exit(2)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    api = args.path == "model-api"
    ddp = None

    if api:
        from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers

        if args.api_single_thread_backward:
            import nerfstudio_thermal_amd as pkg

            pkg.configure_host(single_thread_backward=True)
        cfg, arena, model = build_model(device, mode=args.mode, nerf_samples=args.nerf_samples)
        eng = model.engine
        optimizers = Optimizers(model.get_param_groups(), optimizer_cls=HipFusedAdam if args.api_optimizer == "hip" else torch.optim.Adam)
        if world > 1:
            # the reference's multi-GPU wrap (pipelines/base_pipeline.py:281-283); its reducer all-reduces the gradients over RCCL during backward
            from torch.nn.parallel import DistributedDataParallel as DDP

            ddp = DDP(model, device_ids=[local], find_unused_parameters=True)
    else:
        cfg, arena, eng = build_engine(device, mode=args.mode, nerf_samples=args.nerf_samples)
    rays = args.rays
    broadcast_params(arena)
    torch.manual_seed(rank_seed(42, rank))  # every rank draws its own pixels (scripts/train.py:97)
    cam_t, idx, img, is_th = make_batch(device, rays, seed=rank_seed(42, rank))
    cache = make_image_cache(device)
    # N > 1: the gradient all-reduce (RCCL) is issued per level range of the main table while the backward is still running
    def make_hook(w):
        if args.dp_shard_optimizer:
            from nerfstudio_thermal_amd.parallel import ShardedGradReducer

            return ShardedGradReducer(w, rank) if args.dp_chunks <= 0 else ShardedGradReducer(w, rank, level_chunks=args.dp_chunks)
        kw = {"transport_dtype": torch.bfloat16} if args.dp_bf16 else {}
        if args.dp_chunks < 0:
            return OverlappedGradReducer(w, **kw)
        return OverlappedGradReducer(w, level_chunks=args.dp_chunks, **kw) if args.dp_chunks > 0 else GradAllReducer(w)

    hook = make_hook(world) if (world > 1 and not api) else None  # (the drop-in path exchanges through DistributedDataParallel instead)
    if hook is not None and args.dp_adam_per_range:
        hook.adam_per_range = True
    if args.force_dp and world == 1:
        from nerfstudio_thermal_amd.parallel import free_port

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if not args.force_dp_sum:
            # AVG on the one-rank group: RCCL then runs a real kernel over every exchanged slice, beside the folds -- as N > 1 does.  (With SUM a
            # one-rank group launches nothing: --force-dp-sum measures the schedule without any collective kernel.)
            os.environ["TN_DP_ONE_RANK_AVG"] = "1"
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
        hook = make_hook(1)
        if args.dp_adam_per_range:
            hook.adam_per_range = True
        if args.dp_chunks == 0:
            hook.force = True  # GradAllReducer returns early at world 1: make it issue the collective

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # GradScaler semantics of the reference's train_iteration (engine/trainer.py:470-495): the drop-in path runs torch.amp.GradScaler itself,
    # the fused step its device-side equivalent (non-finite check of every group's gradients, skip / backoff / schedule lag on the device)
    scaler = None
    if not args.no_grad_scaler:
        if api:
            scaler = torch.amp.GradScaler("cuda")
        else:  # (also with Adam per exchanged range / the sharded optimiser: their found_inf flags are agreed on behind the last exchange)
            from nerfstudio_thermal_amd.optim import DeviceGradScaler

            scaler = DeviceGradScaler(device, num_groups=len(arena.optimised_groups))

    if not api:
        # TN_OVERLAP_ADAM=1: the field groups' Adam launch on a side stream beside the next step's pixel + proposal sampling.  Measured
        # (profiles/r03_experiments.md): 1.09 -> 1.67 ms/step -- with a second queue active across the step boundary nearly every kernel of the
        # step runs 1.5-5x longer, whatever the Adam grid -- so it stays off.
        eng.overlap_adam = os.environ.get("TN_OVERLAP_ADAM", "0") == "1"

    def run(step):
        if api:
            return one_step_api(model, optimizers, cam_t, cache, rays, step, scaler, call=ddp)
        return one_step(eng, cam_t, cache, rays, step, hook, scaler)

    # The "render PSNR vs ref" half of the metric, on the untouched synthetic weights the reference golden was rendered with (before any training
    # step changes them -- the schedule guard's timing steps below included --, outside every timed region)
    parity = None
    if rank == 0 and world == 1 and not api and args.mode == "shared" and args.nerf_samples == 48 and not args.no_extras:
        try:
            parity = parity_vs_reference(eng, device)
        except Exception as e:  # noqa: BLE001  (an extra must never cost the headline line)
            parity = {"error": repr(e)}
    # ---- parallel.ScheduleGuard: the overlapped data-parallel schedule against the plain step, measured here, before the warm-up.  A stall of the
    # overlapped schedule switches the run to the simple one IN PROCESS; the JSON says which schedule ran and why.
    dp_info = None
    if hook is not None and not api:
        dp_info = {"schedule": "overlapped" if getattr(hook, "pipelined", False) else "simple", "guard": None,
                   "streams": "main + one side stream (both proposal networks) + RCCL's; d position in line; one communicator, proposal exchange issued last",
                   "one_rank_reduce_op": (None if world > 1 else ("SUM" if args.force_dp_sum else "AVG"))}
        if getattr(hook, "pipelined", False) and not args.no_dp_guard and not args.dp_shard_optimizer and args.mode == "shared":
            from nerfstudio_thermal_amd.parallel import ScheduleGuard

            guard = ScheduleGuard(world)

            def timed(h, n=6, skip=2):
                nonlocal gstep
                ts = []
                for _ in range(n):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    one_step(eng, cam_t, cache, rays, gstep % 8, h, scaler)  # (step < 10: EVERY guard step updates the proposal networks --
                    torch.cuda.synchronize()                                 # the three legs time the same, heaviest, kind of step)
                    ts.append((time.perf_counter() - t0) * 1e3)
                    gstep += 1
                return float(np.median(ts[skip:]))

            gstep = 0
            simple_hook = GradAllReducer(world, force=(world == 1))
            plain_ms = timed(None)        # un-exchanged steps: every rank drifts on its own rays ...
            over_ms = timed(hook)
            simple_ms = timed(simple_hook)
            dec = guard.decide(plain_ms, over_ms, simple_ms)
            guard.resync(arena, grad_scaler=scaler)  # ... and is brought back to rank 0's parameters, Adam moments and loss-scale state here
            dp_info["guard"] = dec
            if dec["schedule"] == "simple":
                hook = simple_hook
                dp_info["schedule"] = "simple (picked by the guard: overlapped step %.2f ms, simple step %.2f ms, plain step %.2f ms)" % (
                    dec["overlapped_ms"], dec["simple_ms"], dec["plain_ms"])
                print("bench.py: " + dp_info["schedule"], file=sys.stderr)

    step = 0
    for _ in range(args.warmup):
        run(step)
        step += 1
    # Everything built so far (torch, the model, the arena views, ctypes structs: ~10^6 tracked objects) moves to the permanent generation:
    # a full pass of Python's cyclic collector over them costs 35-40 ms and used to land inside the timed loop every 50-150 steps
    # (+0.25 ms/step on the fused step, +0.7 ms/step on the data-parallel and drop-in paths, which allocate more containers per step).
    # The collector stays ENABLED: garbage created from here on is still found, the passes just stop re-scanning the long-lived heap.
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    barrier()
    updates = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = run(step)
        updates += int(eng.steps_since_update == 1)  # reset to 0 on an iteration that updates the proposal networks, then step_cb adds 1
        step += 1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    rccl_ranks, per_rank_ms = 1, [dt / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(gathered, t)
        per_rank_ms = [float(g.item()) / args.steps * 1e3 for g in gathered]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        one = torch.ones(1, device=device)
        torch.distributed.all_reduce(one)  # a real collective: the number of ranks RCCL actually joined
        rccl_ranks = int(one.item())
        assert rccl_ranks == torch.distributed.get_world_size() == world, (rccl_ranks, world)
    final_loss = {k: float(v) for k, v in losses.items()}
    assert all(np.isfinite(v) for v in final_loss.values()), final_loss

    # "median of >= 200 steps after skipping 2".  `value` above follows --steps; this is the long-run figure beside it:
    # every step bracketed by HIP events on the launch stream (no host sync inside the loop), median / mean of the per-step device times.
    long_run = None
    if args.long_steps > 0:
        n_long = args.long_steps
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_long + 1)]
        upd_flags = []
        barrier()
        torch.cuda.synchronize()
        tl0 = time.perf_counter()
        evs[0].record()
        for i in range(n_long):
            run(step)
            upd_flags.append(int(eng.steps_since_update == 1))
            step += 1
            evs[i + 1].record()
        torch.cuda.synchronize()
        barrier()
        tl = time.perf_counter() - tl0
        per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n_long)])
        uf = np.array(upd_flags, dtype=bool)
        med = float(np.median(per[2:]))
        if world > 1:
            tm = torch.tensor([med, tl], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
            med, tl = float(tm[0].item()), float(tm[1].item())
        long_run = {"steps": n_long, "median_ms_per_step": med, "median_rays_per_s": world * rays / (med * 1e-3),
                    "mean_ms_per_step": tl / n_long * 1e3, "mean_rays_per_s": world * rays * n_long / tl,
                    "proposal_update_fraction": float(uf.mean()),
                    "median_ms_update_steps": float(np.median(per[uf])) if uf.any() else None,
                    "median_ms_other_steps": float(np.median(per[~uf])) if (~uf).any() else None,
                    "note": "BASELINE.md 3.4 figure (median of per-step HIP-event times, first 2 skipped; max over ranks); `value` follows --steps"}

    if rank == 0:
        # (measured BEFORE the stand-alone launches below, so that the LAST launches of every kernel in a rocprofv3 trace of this command are the
        # stand-alone ones: scripts/rocpd_stats.py --tail 10 then reproduces roofline.avg_launch_ms, the all-launch mean the in-step figure)
        # the same launch pair INSIDE the step (HIP events on the launch stream around the scatter phase, 20 extra steps): what runs beside it
        # (k_field_dpos; on update steps the proposal networks' backward on the side streams) shares the chip with it
        in_step = None
        if not api and world == 1 and hook is None:
            eng.scatter_events = []
            for _ in range(20):
                run(step)
                step += 1
            torch.cuda.synchronize()
            tms = [(e0.elapsed_time(e1), u) for e0, e1, u in eng.scatter_events]
            eng.scatter_events = None
            if tms:
                in_step = {"mean_ms": float(np.mean([t for t, _ in tms])),
                           "mean_ms_update_steps": float(np.mean([t for t, u in tms if u])) if any(u for _, u in tms) else None,
                           "mean_ms_other_steps": float(np.mean([t for t, u in tms if not u])) if any(not u for _, u in tms) else None}
        rows, mfma_rows = kernel_roofline(eng, cam_t, idx, img, is_th)
        if args.ops:
            for name, ms, nbytes in rows:
                print(f"{name:60s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s algorithmic", file=sys.stderr)
        # the dominant HBM-bound launch (pair) = largest measured share of a step: stand-alone time x launches per step (the proposal grids'
        # scatters run on update steps only).  The MFMA-bound kernel has its own object (`roofline.mfma`).
        upd_share = updates / max(args.steps, 1)
        shares = {r[0]: r[1] * (upd_share if "prop" in r[0] and r[0].startswith("scatter") else 1.0) for r in rows}
        name, ms, nbytes = max(rows, key=lambda r: shares[r[0]])
        achieved = nbytes / (ms * 1e-3) / 1e9
        pmc, why = load_pmc() if (rays == 4096 and args.mode == "shared") else (None, "PMC passes cover the 4096-ray shared workload only")
        traffic = None
        if pmc is not None and pmc_lookup(pmc, PMC_KEYS.get(name, ["?"])) is not None:
            traffic = sum(v["traffic_bytes"] for v in pmc_lookup(pmc, PMC_KEYS[name]))
        elif pmc is not None:
            why = f"profiles/r06_pmc.json holds no (single) entry for {PMC_KEYS.get(name)}"
        upd = updates / max(args.steps, 1)
        step_bytes = step_algorithmic_bytes(arena, args.mode, rays, upd, args.nerf_samples)
        step_gbs = step_bytes / (dt / args.steps) / 1e9
        roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms,
                    "avg_launch_ms_in_step": None if in_step is None else in_step["mean_ms"], "in_step": in_step,
                    "in_step_schedule": "phased backward: tn_field_bwd_phase(MLP) / HIP event / SCATTER / HIP event / JOIN in 20 EXTRA steps after the timed "
                                        "region -- the timed `value` runs the one-call backward (tn_render_rays_train_bwd), which cannot take events",
                    "frac_in_step": None if in_step is None else nbytes / (in_step["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "the main grid's scatter entry point = k_seg_bin + k_seg_fold as the field backward calls it (d position comes from "
                            "k_field_dpos); the bin pass is bound by instruction issue, the fold streams its records into double-precision LDS atomics",
                    "dominant_by": {"rule": "stand-alone ms x launches per step at this run's proposal-update fraction", "ms_per_step": shares},
                    # rows whose table is L2-resident (the proposal grids: 5 MB each, 12 MB fetched for 335 MB algorithmic in the PMC passes) are stated
                    # against the L2 rate of the guide, not against HBM -- their algorithmic bytes carry no cache credit and would exceed 1.0 there
                    "all_kernels": {n: ({"ms": m, "GB/s": bts / (m * 1e-3) / 1e9, "bound": "l2", "peak": L2_PEAK_GBS, "frac": bts / (m * 1e-3) / 1e9 / L2_PEAK_GBS}
                                        if n.startswith("k_prop_fwd") else
                                        {"ms": m, "GB/s": bts / (m * 1e-3) / 1e9, "bound": "hbm", "peak": HBM_PEAK_GBS, "frac": bts / (m * 1e-3) / 1e9 / HBM_PEAK_GBS})
                                    for n, m, bts in rows},
                    # the MFMA-bound launches (fp32 MFMA: v_mfma_f32_32x32x2_f32 / 16x16x4_f32): achieved TFLOP/s against the dense fp32 peak
                    "mfma": mfma_rows[1], "mfma_all": mfma_rows,
                    # SURVEY 8d's whole-step figure: (N x bytes_ray + bytes_step) / t_step against the HBM peak
                    "step": {"algorithmic_bytes": step_bytes, "achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS, "proposal_update_fraction": upd}}
        if traffic is None:
            roofline["traffic_unavailable"] = why
        if pmc is not None:
            roofline["traffic_all_kernels"] = {n: sum(v["traffic_bytes"] for v in pmc_lookup(pmc, ks)) for n, ks in PMC_KEYS.items()
                                               if pmc_lookup(pmc, ks) is not None}
            roofline["mfma_busy_frac"] = {k.split(" ")[0]: v["mfma_busy_frac"] for k, v in pmc.items()
                                          if k.split(" ")[0] in ("k_field_mlp_fwd<true>", "k_field_bwd_fused<false>") and "mfma_busy_frac" in v}
            roofline["pmc_source"] = "profiles/r06_pmc.json (read bytes = FETCH_SIZE x the calibrated factor of the kernel's read shape, scripts/pmc_summary.py)"
        result = {
            "metric": "train rays/sec (4096-ray batch, 96 samples/ray)",
            "value": world * rays * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"thermal-nerfacto density_mode={args.mode} train step (pixel sampling+raygen+fwd+losses+bwd+allreduce+Adam), {rays} rays/GPU, "
                                   f"256/96 proposal + {args.nerf_samples} field samples, hash 16x2^19x2 + 2x(5x2^17x2), 8 cameras (4 RGB + 4 thermal)",
                       "rays_per_gpu": rays, "parallelism": f"dp{world}", "path": args.path + (f" ({args.api_optimizer} Adam" + (", single-thread backward" if args.api_single_thread_backward else "") + ")" if api else ""),
                       "grad_scaler": (None if scaler is None else ("torch.amp.GradScaler + autocast" if api else "device-side (optim.DeviceGradScaler)")),
                       "final_loss": final_loss},
            "rccl_ranks": rccl_ranks,
            "per_rank_ms_per_step": per_rank_ms,
            "dp": dp_info,
            "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES"),
            "long_run": long_run,
            "roofline": roofline,
        }
        if parity is not None:
            result["parity"] = parity
        default_run = world == 1 and not api and hook is None and args.mode == "shared" and rays == RAYS_PER_GPU and args.nerf_samples == 48
        if default_run and not args.no_extras:
            # after the timed region, outside `value`: the render parity figure BASELINE.json's metric names, and the configs the driver never runs
            result["extra"] = {}
            for key, kw in (("separate_8192", dict(mode="separate", rays=8192, nerf_samples=48, path="fused", steps=12, warmup=4)),
                            ("nerf_samples_96", dict(mode="shared", rays=4096, nerf_samples=96, path="fused", steps=20, warmup=6)),
                            ("model_api_amp", dict(mode="shared", rays=4096, nerf_samples=48, path="model-api", steps=20, warmup=6)),
                            ("fused_trainer", dict(mode="shared", rays=4096, nerf_samples=48, path="fused-trainer", steps=20, warmup=6))):
                try:
                    result["extra"][key] = extra_leg(device, **kw)
                except Exception as e:  # noqa: BLE001
                    result["extra"][key] = {"error": repr(e)}
                torch.cuda.empty_cache()
            try:  # opt-in experiment (TN_HEAD_BF16X3=1): its own timing and parity block, never the headline
                result["extra"]["head_bf16x3"] = head_bf16x3_leg(device)
            except Exception as e:  # noqa: BLE001
                result["extra"]["head_bf16x3"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            try:  # the eval / render side of the metric: full images through get_outputs_for_camera and through tn_render_rays_eval
                result["extra"]["eval_render"] = eval_render_leg(device, profile_path=os.environ.get("TN_EVAL_PROFILE"))
            except Exception as e:  # noqa: BLE001
                result["extra"]["eval_render"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            cores = args.cpu_threads or min(os.cpu_count() or 1, 16)
            v, t = cpu_baseline(rays, args.cpu_steps if args.mode == "shared" else max(2, args.cpu_steps // 2), cores, args.mode)
            result["cpu_baseline"] = {"value": v, "unit": "rays/s", "cores": cores, "kind": "port", "host_cores": os.cpu_count(),
                                      "threads_note": "min(host cores, 16) torch threads: more threads make this op mix SLOWER on the GPU box's 256-thread host "
                                                      "(measured at 1024 rays: 8 -> 2427, 16 -> 2286, 32 -> 1906, 64 -> 1025, 256 -> 24 rays/s); --cpu-threads overrides",
                                      "sample": f"{args.cpu_steps if args.mode == 'shared' else max(2, args.cpu_steps // 2)} steps (after 1 warm-up) of the same "
                                                f"{rays}-ray {args.mode}-density train step (fwd+losses+bwd+Adam) by the oracle, median {t:.2f} s/step"}
            if args.mode == "shared":  # comparability with the figures of SURVEY 6 / BASELINE.md: 8 threads, and 1 thread on a quarter batch
                v8, t8 = cpu_baseline(rays, 2, min(8, os.cpu_count() or 1), args.mode)
                v1, t1 = cpu_baseline(rays // 4, 1, 1, args.mode)
                result["cpu_baseline"]["other_thread_counts"] = {"8": {"value": v8, "s_per_step": t8, "steps": 2},
                                                                 "1": {"value": v1, "s_per_step": t1, "steps": 1, "rays": rays // 4}}
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
