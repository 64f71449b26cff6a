"""ThermalNerfactoModel for MI355X: the reference's Model API (models/base_model.py:57-263, models/nerfacto.py:136-447,
models/thermal_nerfacto.py:67-564) over the HIP hot path.

 * construction: `ThermalNerfactoModelConfig(...).setup(scene_box=, num_train_data=, metadata={"is_thermal": [...]}, device=)`
 * state_dict keys / shapes are the reference's (checkpoints round-trip), including the aliased proposal hash tables and the
   int64 buffers; every Parameter is a view into one flat arena (arena.py)
 * forward(ray_bundle) / get_outputs / get_metrics_dict / get_loss_dict / get_param_groups / get_training_callbacks /
   get_outputs_for_camera_ray_bundle keep their signatures; in train mode the outputs carry an autograd edge (one fused
   torch.autograd.Function for the whole render), so `sum(loss_dict.values()).backward()` fills `param.grad` as the Trainer expects
 * `train_iteration(ray_bundle, batch, step)` is the fast path: forward + every loss + backward + Adam in one fused sequence of
   kernel launches with no autograd tape (what bench.py times)
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import autograd_ops as F
from . import ops
from .arena import ParamArena
from .config import ThermalNerfactoModelConfig
from .engine import RenderEngine
from .model_components import (
    AccumulationRenderer,
    CameraOptimizer,
    DepthRenderer,
    HashMLPDensityField,
    NearFarCollider,
    ProposalNetworkSampler,
    RGBRenderer,
    RGBTRenderer,
    ThermalNerfactoField,
)
from .rays import LazyRaySamples, RayBundle


class TrainingCallbackLocation(Enum):
    """engine/callbacks.py:44-56."""

    BEFORE_TRAIN_ITERATION = auto()
    AFTER_TRAIN_ITERATION = auto()
    AFTER_TRAIN = auto()


@dataclass
class TrainingCallback:
    """engine/callbacks.py:59-115 (the subset the model registers)."""

    where_to_run: List[TrainingCallbackLocation]
    func: Callable
    update_every_num_iters: Optional[int] = None

    def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
        if location in self.where_to_run and (self.update_every_num_iters is None or step % self.update_every_num_iters == 0):
            self.func(step)


@dataclass
class SceneBox:
    """data/scene_box.py:29-80 (the aabb buffer is all the path needs)."""

    aabb: Tensor


def _psnr(pred: Tensor, gt: Tensor) -> Tensor:
    """PeakSignalNoiseRatio(data_range=1.0)."""
    return -10.0 * torch.log10(torch.mean((pred - gt) ** 2))


def _ssim(pred: Tensor, gt: Tensor, kernel_size: int = 11, sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> Tensor:
    """Structural similarity (Wang et al. 2004) of [1,C,H,W] images with a Gaussian window; reflect padding, borders cropped, mean over
    channels and pixels; data range = max - min over both images (torchmetrics' default when data_range is None)."""
    C = pred.shape[1]
    ax = torch.arange(kernel_size, dtype=pred.dtype, device=pred.device) - (kernel_size - 1) / 2.0
    g1 = torch.exp(-(ax / sigma) ** 2 / 2.0)
    g1 = g1 / g1.sum()
    rng = torch.maximum(pred.max() - pred.min(), gt.max() - gt.min())
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    pad = (kernel_size - 1) // 2
    p = torch.nn.functional.pad(pred, (pad, pad, pad, pad), mode="reflect")
    t = torch.nn.functional.pad(gt, (pad, pad, pad, pad), mode="reflect")
    # the window is separable (outer product of g1 with itself): two 1-D passes over unfolded views -- plain element-wise kernels.  (A grouped
    # conv2d goes through MIOpen, whose first call per image shape searches / compiles kernels for seconds: most of an evaluation pass of a few
    # images was that.)
    f = lambda x: (((x.unfold(2, kernel_size, 1) * g1).sum(-1)).unfold(3, kernel_size, 1) * g1).sum(-1)  # noqa: E731
    mu_p, mu_t = f(p), f(t)
    s_pp, s_tt, s_pt = f(p * p) - mu_p * mu_p, f(t * t) - mu_t * mu_t, f(p * t) - mu_p * mu_t
    ssim = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    return ssim[..., pad:-pad, pad:-pad].mean() if ssim.shape[-1] > 2 * pad and ssim.shape[-2] > 2 * pad else ssim.mean()


def rgb_to_rgbt_image(image: Tensor, is_thermal: Tensor) -> Tensor:
    """utils/rgbt_utils.py:6-32."""
    rgbt = torch.zeros(image.shape[:-1] + (4,), device=image.device)
    rgbt[..., :3] = image * (1 - is_thermal)[:, None]
    rgbt[..., 3] = image[..., 0] * is_thermal
    return rgbt


class _RenderFn(torch.autograd.Function):
    """The whole train-mode render as ONE autograd node: forward = engine.get_outputs, backward = the backward kernel sequence.
    Tensor outputs (per branch): comp [N,C], density [N,S2,1], weights of the 3 levels [N,S,1]; separate mode adds density2 / density2_thermal."""

    @staticmethod
    def forward(ctx, model, holder, origins, directions, cam, jitters, jitters_thermal, *params):
        eng: RenderEngine = model.engine
        # The backward kernels ACCUMULATE into the arena's gradient buffer, and autograd usually leaves `param.grad` aliased to it (the views
        # returned by backward are adopted, not copied).  A forward that finds such aliased gradients still alive is a further micro-step of
        # gradient accumulation (engine/trainer.py:464-477: several forward/backward passes before one optimiser step): the buffer must then
        # keep what it holds.  Otherwise (the Trainer zeroed the gradients: set_to_none) this is a fresh step and the buffer is cleared.
        ctx.accumulating = model._grads_alias_arena()
        if not ctx.accumulating:
            eng.arena.zero_grad()
        out, branches = eng.get_outputs(origins, directions, cam, True, jitters, jitters_thermal)
        # The node keeps the engine, the arena's parameters and the branches (plain tensors) -- NOT the model and NOT the output dict: the
        # caller fills that dict with this node's own outputs, and a node that reaches its outputs (directly or through the model) is a
        # reference cycle that only the cyclic garbage collector frees: ~30 MB of activations per iteration piled up until it ran.
        ctx.eng, ctx.branches, ctx.cam = eng, branches, cam
        # outputs nobody differentiated arrive as None in backward, not as zero tensors: an iteration whose proposal networks ran without
        # gradients then takes the one-call backward below instead of three fills and the launch-by-launch path
        ctx.set_materialize_grads(False)
        ctx.names, ctx.params = model._param_names, params
        ctx.has_cross = bool(eng.separate and "density2" in out)
        tensors = []
        for sfx, br in branches.items():
            tensors += [br.comp, out[f"density{sfx}"]] + out[f"weights_list{sfx}"]
        if ctx.has_cross:
            tensors += [out["density2"], out["density2_thermal"]]
        holder.append((out, branches))
        # hand out ALIASES: autograd makes the returned tensor objects point at this node, and `branches` (kept by the node) holds br.comp
        # itself -- returning that object would close the cycle node -> branches -> comp -> node again
        return tuple(t.detach() for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        eng: RenderEngine = ctx.eng
        branches, cam = ctx.branches, ctx.cam
        arena = eng.arena
        arena.grads_clean = False  # this node (and autograd, through the aliased .grad views) writes into the arena's gradient buffer
        eng._set_grad_zero(False)  # (several backward passes may add into one arena: the scatters must add, not store)
        dev = eng.device
        it = iter(grads)
        N = cam.shape[0]
        per = {}
        for sfx, br in branches.items():
            g_comp, g_dens = next(it), next(it)
            g_w = [next(it) for _ in range(3)]
            per[sfx] = (g_comp, g_dens, g_w)
        g_d2 = g_d2t = None
        if ctx.has_cross:
            g_d2, g_d2t = next(it), next(it)
        z = lambda ref: torch.zeros_like(ref)  # noqa: E731
        # gradient accumulation with a parameter whose .grad is NOT the arena view (autograd summed two gradient sources into a buffer of its
        # own, e.g. the pose: render + regulariser): its slice of the arena still holds the previous micro-steps, so this backward must start
        # from zero there and hand out only its own contribution
        detached = {}
        if ctx.accumulating:
            params = dict(zip(ctx.names, ctx.params))
            for n in ctx.names:
                g = params[n].grad
                if g is not None and g.data_ptr() != arena.grad_ptr(n):
                    view = arena.grad_view(n)
                    detached[n] = view.clone()
                    view.zero_()
        d_od = {}
        for sfx, br in branches.items():
            g_comp, g_dens, g_w = per[sfx]
            fld = eng.field_thermal if sfx else eng.field
            props = eng.props_thermal if sfx else eng.props
            pose = eng.pose_thermal if sfx else eng.pose
            lv = br.levels
            d_o = torch.zeros((N, 3), device=dev) if pose is not None else None
            d_d = torch.zeros((N, 3), device=dev) if pose is not None else None
            dw2 = (g_w[2][..., 0].contiguous() if g_w[2] is not None else z(lv[2].weights))  # read only (tn_render_bwd)
            both, neither = g_w[0] is not None and g_w[1] is not None, g_w[0] is None and g_w[1] is None
            if br.fwd_buf is not None and (neither or (both and br.prop_grad)):
                # the whole backward of the branch as ONE call of the C ABI (tn_render_rays_train_bwd), exactly as the fused step issues it
                # (engine.loss_and_backward): renderer backward, the density loss's own gradient added, both proposal networks on the
                # library's companion streams, field backward with d position and table scatter
                pg = both and br.prop_grad
                ops.render_rays_train_bwd(props, fld, br.fwd_buf, br.origins, br.directions, cam, eng.counts,
                                          (g_comp.contiguous() if g_comp is not None else z(br.comp)),
                                          [g_w[0][..., 0].contiguous() if pg else None, g_w[1][..., 0].contiguous() if pg else None, dw2],
                                          g_dens[..., 0].contiguous() if g_dens is not None else None, d_o, d_d,
                                          tag="main", side_tags=("side0" + sfx, "side1" + sfx), prop_enc_saved=br.prop_enc_saved)
                d_od[sfx] = (d_o, d_d)
                continue
            d_rgb, d_dens = ops.render_bwd(lv[2].e_bins, lv[2].density, br.rgb_samples, lv[2].weights,
                                           (g_comp.contiguous() if g_comp is not None else z(br.comp)), dw2)
            if g_dens is not None:
                d_dens += g_dens[..., 0]
            # same schedule as the fused step (engine.loss_and_backward): the level-0 proposal network (2/3 of the proposal work) on the side
            # stream beside the main field's backward, the level-1 network on a second one
            side = None
            if br.prop_grad and g_w[0] is not None:
                g0 = g_w[0][..., 0].contiguous()
                side = eng._side_stream()
                side.wait_stream(torch.cuda.current_stream())  # g0, d_o, d_d are produced / zeroed on the main stream
                with torch.cuda.stream(side):
                    dd = ops.weights_bwd(lv[0].e_bins, lv[0].density, lv[0].weights, g0)
                    ops.prop_density_bwd(props[0], br.origins, br.directions, lv[0].e_bins, dd, d_o, d_d, tag="side0")
            side1 = None
            if br.prop_grad and g_w[1] is not None:
                g1 = g_w[1][..., 0].contiguous()
                side1 = eng._side_stream(1)
                side1.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side1):
                    dd = ops.weights_bwd(lv[1].e_bins, lv[1].density, lv[1].weights, g1)
                    ops.prop_density_bwd(props[1], br.origins, br.directions, lv[1].e_bins, dd, d_o, d_d, tag="side1")
            ops.field_bwd(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d)
            for st in (side, side1):
                if st is not None:
                    torch.cuda.current_stream().wait_stream(st)
            d_od[sfx] = (d_o, d_d)
        if g_d2 is not None or g_d2t is not None:
            b, bt = branches[""], branches["_thermal"]
            if g_d2 is not None:
                ops.field_bwd(eng.field, bt.origins, bt.directions, cam, bt.levels[-1].e_bins, g_d2[..., 0].contiguous(), None, *d_od["_thermal"],
                              tag="cross")
            if g_d2t is not None:
                ops.field_bwd(eng.field_thermal, b.origins, b.directions, cam, b.levels[-1].e_bins, g_d2t[..., 0].contiguous(), None, *d_od[""],
                              tag="cross")
        for sfx, br in branches.items():
            pose = eng.pose_thermal if sfx else eng.pose
            if pose is None:
                continue
            ops.pose_apply_bwd(pose, eng.frozen_thermal if sfx else eng.frozen_rgb, cam, br.directions_in, *d_od[sfx],
                               eng.pose_thermal_grad if sfx else eng.pose_grad)
        # Hand the arena's gradient views to autograd as the parameters' gradients.  None (= "no gradient", as under the reference's no_grad)
        # for every parameter this step did not differentiate: the proposal networks on iterations where the sampler did not update them
        # (ray_samplers.py:591,605-610) and the thermal twins that shared mode never evaluates -- torch.optim.Adam skips a parameter whose
        # .grad is None, whereas a zero gradient would advance its step count and let it coast on its momentum.  None also for a parameter
        # whose .grad already IS the arena view (gradient accumulation): the kernels have added into it in place.
        key = tuple((sfx, bool(br.prop_grad)) for sfx, br in branches.items())
        cache = arena.__dict__.setdefault("_live_index", {})
        hit = cache.get(key)
        if hit is None or hit[0] is not ctx.names:  # (positions in ctx.names of the parameters this kind of iteration differentiates)
            live = set()
            for sfx, pgrad in key:
                live.update(arena.group_keys["fields_thermal" if sfx else "fields"])
                live.update(arena.group_keys["camera_opt_thermal" if sfx else "camera_opt"])
                if pgrad:
                    live.update(arena.group_keys["proposal_networks_thermal" if sfx else "proposal_networks"])
            hit = cache[key] = (ctx.names, [i for i, n in enumerate(ctx.names) if n in live])
        names, plist = ctx.names, ctx.params
        pg = [None] * len(names)
        ptrs = arena.grad_ptrs()
        for i in hit[1]:
            n = names[i]
            if n in detached:
                view = arena.grad_view(n)
                pg[i] = view.clone()
                view.add_(detached.pop(n))  # the arena keeps the running total
                continue
            g = plist[i].grad
            if g is None or g.data_ptr() != ptrs[n]:
                pg[i] = arena.grad_view(n)  # (a FRESH view object per step: autograd adopts a gradient it holds the only reference to, and copies one it does not)
        for n, prev in detached.items():  # (not differentiated this iteration: its slice just gets the previous micro-steps back)
            arena.grad_view(n).add_(prev)
        return (None, None, None, None, None, None, None, *pg)


class ThermalNerfactoModel(nn.Module):
    config: ThermalNerfactoModelConfig

    def __init__(self, config: ThermalNerfactoModelConfig, scene_box, num_train_data: int, **kwargs) -> None:
        super().__init__()
        config.validate_for_hip()
        self.config = config
        self.scene_box = scene_box
        self.render_aabb = None
        self.num_train_data = num_train_data
        self.kwargs = kwargs
        dev = kwargs.get("device", "cuda")
        self._device = torch.device(dev)
        if self._device.type != "cuda":
            raise RuntimeError("ThermalNerfactoModel(implementation='hip') needs a HIP device: there is no CPU fallback on this path")
        ops._lib.load()
        self.collider = None
        self.populate_modules()
        self.callbacks = None
        self.device_indicator_param = nn.Parameter(torch.empty(0, device=self._device))
        self.step = 0

    @property
    def device(self):
        return self.device_indicator_param.device

    # ------------------------------------------------------------------------------------------------ construction
    def populate_modules(self):
        cfg = self.config
        is_thermal = list(self.kwargs["metadata"]["is_thermal"])
        self.arena = ParamArena(cfg, self.num_train_data, self._device)
        self._init_parameters()
        self.engine = RenderEngine(cfg, self.arena, self.num_train_data, is_thermal)
        eng = self.engine
        aabb = torch.as_tensor(self.scene_box.aabb, dtype=torch.float32).to(self._device)
        self.output_suffixes = ("", "_thermal") if cfg.density_mode == "separate" else ("",)
        self._params: Dict[str, nn.Parameter] = {n: nn.Parameter(self.arena.view(n)) for n in self.arena.names()}

        def P(name):
            return self._params[name]

        def mk_field(prefix, fld):
            names = {
                "emb": "embedding_appearance.embedding.weight", "table": "mlp_base.model.0.hash_table",
                "w0": "mlp_base.model.1.layers.0.weight", "b0": "mlp_base.model.1.layers.0.bias",
                "w1": "mlp_base.model.1.layers.1.weight", "b1": "mlp_base.model.1.layers.1.bias",
                "hw0": "mlp_head.layers.0.weight", "hb0": "mlp_head.layers.0.bias", "hw1": "mlp_head.layers.1.weight",
                "hb1": "mlp_head.layers.1.bias", "hw2": "mlp_head.layers.2.weight", "hb2": "mlp_head.layers.2.bias",
            }
            return ThermalNerfactoField(aabb, fld, cfg.max_res, {k: P(f"{prefix}.{v}") for k, v in names.items()}, names,
                                        cfg.use_average_appearance_embedding)

        def mk_props(prefix, nets):
            mods = nn.ModuleList()
            for i, net in enumerate(nets):
                a = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
                ps = {"table": P(f"{prefix}.{i}.mlp_base.0.hash_table"), "w0": P(f"{prefix}.{i}.mlp_base.1.layers.0.weight"),
                      "b0": P(f"{prefix}.{i}.mlp_base.1.layers.0.bias"), "w1": P(f"{prefix}.{i}.mlp_base.1.layers.1.weight"),
                      "b1": P(f"{prefix}.{i}.mlp_base.1.layers.1.bias")}
                mods.append(HashMLPDensityField(aabb, net, a["max_res"], ps))
            return mods

        self.field = mk_field("field", eng.field)
        if cfg.density_mode == "separate":
            self.field_thermal = mk_field("field_thermal", eng.field_thermal)
        rgb_frozen = torch.tensor([i for i, x in enumerate(is_thermal) if x != 0], dtype=torch.long)
        th_frozen = torch.tensor([i for i, x in enumerate(is_thermal) if x == 0], dtype=torch.long)
        self.camera_optimizer = CameraOptimizer(cfg.camera_optimizer, self.num_train_data, self._device, rgb_frozen,
                                                pose_param=self._params.get("camera_optimizer.pose_adjustment"))
        self.camera_optimizer_thermal = CameraOptimizer(cfg.camera_optimizer_thermal, self.num_train_data, self._device, th_frozen,
                                                        pose_param=self._params.get("camera_optimizer_thermal.pose_adjustment"), suffix="_thermal")
        self.shared_camera_optimizer = CameraOptimizer(cfg.shared_camera_optimizer, self.num_train_data, self._device, rgb_frozen, suffix="_shared")
        self.shared_camera_optimizer_thermal = CameraOptimizer(cfg.shared_camera_optimizer_thermal, self.num_train_data, self._device, th_frozen,
                                                               suffix="_shared_thermal")
        self.proposal_networks = mk_props("proposal_networks", eng.props)
        self.density_fns = [n.density_fn for n in self.proposal_networks]
        self.proposal_networks_thermal = mk_props("proposal_networks_thermal", eng.props_thermal)
        self.density_fns_thermal = [n.density_fn for n in self.proposal_networks_thermal]

        def update_schedule(step):
            return np.clip(np.interp(step, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]), 1, cfg.proposal_update_every)

        mk_sampler = lambda: ProposalNetworkSampler(  # noqa: E731
            num_nerf_samples_per_ray=cfg.num_nerf_samples_per_ray, num_proposal_samples_per_ray=cfg.num_proposal_samples_per_ray,
            num_proposal_network_iterations=cfg.num_proposal_iterations, single_jitter=cfg.use_single_jitter, update_sched=update_schedule)
        self.proposal_sampler = mk_sampler()
        self.proposal_sampler_thermal = mk_sampler()
        self.collider = NearFarCollider(near_plane=cfg.near_plane, far_plane=cfg.far_plane)
        self.renderer_rgb = RGBRenderer(background_color=cfg.background_color)
        self.renderer_rgbt = RGBTRenderer(background_color=cfg.background_color)
        self.renderer_thermal = RGBRenderer(background_color=cfg.background_color, num_channels=1)
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer(method="median")
        self.renderer_expected_depth = DepthRenderer(method="expected")
        self.rgb_loss = nn.MSELoss()
        self.density_loss = nn.L1Loss()
        self.psnr = _psnr
        # parameters in arena order: what _RenderFn receives / returns gradients for
        # (only the optimised groups: the thermal twins that shared mode never evaluates would be ~30 more inputs of the autograd node, each
        # of which costs host time per iteration and could only ever receive None)
        opt = {n for g in self.arena.optimised_groups for n in self.arena.group_keys[g]}
        self._param_names = [n for n in self.arena.names() if n in opt]
        self._param_list = [self._params[n] for n in self._param_names]

    def _init_parameters(self) -> None:
        """The reference's initialisers: hash tables U(-1,1)*1e-3 (field_components/encodings.py:377-379), nn.Linear default
        (kaiming-uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)) for weight and bias), nn.Embedding N(0,1), pose_adjustment zeros."""
        g = torch.Generator(device="cpu")
        g.manual_seed(int(self.kwargs.get("seed", 0)) if isinstance(self.kwargs.get("seed", 0), int) else 0)
        a = self.arena
        fan_in = {}
        for name, (_, shape) in a.layout.items():
            if name.endswith(".weight") and "embedding" not in name:
                fan_in[name[: -len("weight")]] = shape[1]
        for name, (_, shape) in a.layout.items():
            if name.endswith("hash_table"):
                t = (torch.rand(shape, generator=g) * 2 - 1) * 1e-3
            elif name.endswith("embedding.weight"):
                t = torch.randn(shape, generator=g)
            elif name.endswith(".weight"):
                b = 1.0 / np.sqrt(shape[1])
                t = (torch.rand(shape, generator=g) * 2 - 1) * b
            elif name.endswith(".bias"):
                b = 1.0 / np.sqrt(fan_in[name[: -len("bias")]])
                t = (torch.rand(shape, generator=g) * 2 - 1) * b
            else:  # pose_adjustment
                t = torch.zeros(shape)
            a.view(name).copy_(t.to(a.device))

    # ------------------------------------------------------------------------------------------------ trainer hooks
    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """models/nerfacto.py:256-261 + models/thermal_nerfacto.py:390-401."""
        a = self.arena
        groups = {g: [self._params[k] for k in a.group_keys[g]] for g in a.optimised_groups}
        return groups

    def get_training_callbacks(self, training_callback_attributes=None) -> List[TrainingCallback]:
        """models/nerfacto.py:263-297 (the thermal sampler's callbacks are only registered with use_proposal_thermal_weight_anneal)."""
        cbs: List[TrainingCallback] = []
        if self.config.use_proposal_weight_anneal:
            def set_anneal(step):
                self.step = step
                self.engine.set_anneal_for_step(step)
                self.proposal_sampler.set_anneal(self.engine.anneal)

            def step_cb(step):
                self.engine.step_cb(step)
                self.proposal_sampler.step_cb(step)

            cbs.append(TrainingCallback([TrainingCallbackLocation.BEFORE_TRAIN_ITERATION], set_anneal, update_every_num_iters=1))
            cbs.append(TrainingCallback([TrainingCallbackLocation.AFTER_TRAIN_ITERATION], step_cb, update_every_num_iters=1))
        return cbs

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, ray_bundle: RayBundle) -> Dict[str, Any]:
        """models/base_model.py:132-143 (the collider's near/far constants are applied inside the engine; the bundle still receives them)."""
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        return self.get_outputs(ray_bundle)

    def get_outputs(self, ray_bundle: RayBundle, jitters=None, jitters_thermal=None) -> Dict[str, Any]:
        """models/thermal_nerfacto.py:403-489."""
        o = ray_bundle.origins.contiguous()
        d = ray_bundle.directions.contiguous()
        cam = ray_bundle.camera_indices.reshape(-1).contiguous()
        eng = self.engine
        grad_mode = self.training and torch.is_grad_enabled()
        if not grad_mode:
            out, branches = eng.get_outputs(o, d, cam, self.training, jitters, jitters_thermal)
        else:
            holder: list = []
            flat = _RenderFn.apply(self, holder, o, d, cam, jitters, jitters_thermal, *self._param_list)
            out, branches = holder.pop()
            it = iter(flat)
            for sfx in branches:
                out[f"rgb{sfx}"] = next(it)
                out[f"density{sfx}"] = next(it)
                out[f"weights_list{sfx}"] = [next(it) for _ in range(3)]
            if eng.separate and "density2" in out:
                out["density2"], out["density2_thermal"] = next(it), next(it)
            if not eng.separate:
                rgbt = out["rgb"]
                out["rgbt"], out["rgb"], out["rgb_thermal"] = rgbt, rgbt[..., :3], rgbt[..., 3:]
        if self.training:
            nears, fars = eng._nears_fars(o.shape[0], True)
            for sfx, br in branches.items():
                out[f"_prop_grad{sfx}"] = br.prop_grad  # the sampler ran this iteration's proposal networks with gradients (ray_samplers.py:591)

                def bundle(br=br, cache=[]):  # built when a level is first used as a real RaySamples (the training loop reads the bins only)
                    if not cache:
                        cache.append(RayBundle(origins=br.origins, directions=br.directions, pixel_area=ray_bundle.pixel_area,
                                               camera_indices=ray_bundle.camera_indices, nears=nears[:, None], fars=fars[:, None]))
                    return cache[0]

                out[f"ray_samples_list{sfx}"] = [LazyRaySamples(bundle, L.s_bins, L.e_bins, nears, fars) for L in br.levels]
        return out

    def _grads_alias_arena(self) -> bool:
        ptrs = self.arena.grad_ptrs()
        for n, p in zip(self._param_names, self._param_list):
            g = p.grad
            if g is not None and g.data_ptr() == ptrs[n]:
                return True
        return False

    def _loss_terms(self, outputs, batch):
        """Every loss term of the iteration as ONE autograd node (autograd_ops.TrainLosses: the fused step's launches), shared by
        get_metrics_dict (PSNRs, distortion) and get_loss_dict.  Returns the unbound loss vector (see TrainLosses)."""
        key = (id(outputs), id(batch))
        cached = self.__dict__.get("_loss_cache")
        if cached is not None and cached[0] == key and cached[1] is outputs:
            return cached[2]
        c = self.config
        sep = c.density_mode == "separate"
        is_th = batch["is_thermal"].to(self.device).float().contiguous()
        img = batch["image"].to(self.device)
        img = (img if img.shape[-1] == 3 else img[..., :3]).contiguous()
        branches, ts = [], []
        for s_ in self.output_suffixes:
            ws = outputs[f"weights_list{s_}"]
            pg = bool(outputs.get(f"_prop_grad{s_}", self.training and ws[0].requires_grad))
            branches.append((s_, [r.s_bins for r in outputs[f"ray_samples_list{s_}"]], pg))
            if sep:
                comp = outputs["rgb_thermal"] if s_ else outputs["rgb"]
            else:
                comp = outputs["rgbt"] if "rgbt" in outputs else torch.cat([outputs["rgb"], outputs["rgb_thermal"]], -1)
            ts += [comp, *ws]
        dw = None
        if sep and c.density_loss_mult > 0 and "density2" in outputs:
            dw = (c.density_loss_mult, c.rgb_density_loss_mult * c.density_loss_mult)
            ts += [outputs["density2"], outputs["density_thermal"], outputs["density"], outputs["density2_thermal"]]
        regs = []
        for co in ((self.camera_optimizer, self.camera_optimizer_thermal) if sep else (self.camera_optimizer,)):
            if co.config.mode != "off":
                regs.append((co.config.trans_l2_penalty, co.config.rot_l2_penalty, co.config.penalty_scale))
                ts.append(co.pose_adjustment)
        spec = F.LossSpec(sep, c.thermal_loss_mult, c.tv_pixel_loss_mult, c.cross_channel_loss_mult, c.distortion_loss_mult, c.interlevel_loss_mult,
                          branches, dw, regs)
        terms = F.TrainLosses.apply(spec, img, is_th, *ts)
        self._loss_cache = (key, outputs, terms)
        return terms

    def get_metrics_dict(self, outputs, batch) -> Dict[str, Any]:
        """models/thermal_nerfacto.py:253-282.  In training the PSNRs come out of the pixel-loss kernel's sums (mean squared error over the RGB
        rays / the thermal rays = the masked-mean losses rescaled by the ray counts): no boolean indexing, which would synchronise the device
        every iteration; the distortion metric is the distortion term of the loss node divided by its multiplier."""
        m: Dict[str, Any] = {}
        c = self.config
        if self.training:
            L = self._loss_terms(outputs, batch)
            m["psnr_rgb"], m["psnr_thermal"] = L[16], L[17]
            nsfx = len(self.output_suffixes)
            if c.distortion_loss_mult > 0:
                m["distortion"] = L[9] / (c.distortion_loss_mult * nsfx)
            else:
                m["distortion"] = 0
                for s in self.output_suffixes:
                    m["distortion"] = m["distortion"] + F.DistortionLoss.apply(outputs[f"weights_list{s}"][-1], outputs[f"ray_samples_list{s}"][-1].s_bins)
        else:
            rgb_l, th_l, _, _, n_rgb = self._pixel_terms(outputs, batch)
            with torch.no_grad():
                n = float(outputs["rgb"].shape[0])
                m["psnr_rgb"] = -10.0 * torch.log10(rgb_l.detach() * n / n_rgb)
                m["psnr_thermal"] = -10.0 * torch.log10(th_l.detach() * (n / c.thermal_loss_mult) / (n - n_rgb))
        if self.training:  # the pose norms come out of the loss node's metrics launch (same order as the pose parameters it was given)
            k = 18
            for co in ((self.camera_optimizer, self.camera_optimizer_thermal) if c.density_mode == "separate" else (self.camera_optimizer,)):
                if co.config.mode != "off":
                    m[f"camera_opt_translation{co.suffix}"], m[f"camera_opt_rotation{co.suffix}"] = L[k], L[k + 1]
                    k += 2
        else:
            self.camera_optimizer.get_metrics_dict(m)
            if c.density_mode == "separate":
                self.camera_optimizer_thermal.get_metrics_dict(m)
        return m

    def _pixel_terms(self, outputs, batch):
        """Eval mode: the four pixel losses + the RGB-ray count, one launch shared by get_metrics_dict (PSNRs) and get_loss_dict."""
        key = (id(outputs), id(batch))
        cached = self.__dict__.get("_pixel_cache")
        if cached is not None and cached[0] == key and cached[1] is outputs:
            return cached[2]
        c = self.config
        is_th = batch["is_thermal"].to(self.device).float()
        img = batch["image"].to(self.device)[..., :3].contiguous()
        terms = F.PixelLosses.apply(outputs["rgb"], outputs["rgb_thermal"], img, is_th, c.thermal_loss_mult, c.tv_pixel_loss_mult, c.cross_channel_loss_mult)
        self._pixel_cache = (key, outputs, terms)
        return terms

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, Tensor]:
        """models/thermal_nerfacto.py:284-388.  Training: every term is an output of the one loss node (multipliers folded into the kernels,
        the density cross terms with the reference's detach asymmetry).  Eval: the pixel terms (+ the density loss) through their own kernels."""
        c = self.config
        ld: Dict[str, Any] = {}
        sep_loss = c.density_mode == "separate" and c.density_loss_mult > 0
        if self.training:
            assert metrics_dict is not None and "distortion" in metrics_dict
            L = self._loss_terms(outputs, batch)
            ld["rgb_loss"], ld["thermal_loss"] = L[0], L[1]
            if sep_loss and "density2" in outputs:
                ld["density_loss"] = L[10]
            if c.tv_pixel_loss_mult > 0:
                ld["tv_pixel_loss"] = L[2]
            if c.cross_channel_loss_mult > 0:
                ld["cross_channel_loss"] = L[3]
            ld["interlevel_loss"] = L[8]
            ld["distortion_loss"] = L[9]
            k = 11
            for co in ((self.camera_optimizer, self.camera_optimizer_thermal) if c.density_mode == "separate" else (self.camera_optimizer,)):
                if co.config.mode != "off":
                    ld[f"camera_opt_regularizer{co.suffix}"] = L[k]
                    k += 1
            return ld
        rgb_l, th_l, tv_l, cross_l, _ = self._pixel_terms(outputs, batch)
        ld["rgb_loss"], ld["thermal_loss"] = rgb_l, th_l
        if sep_loss and "density2" in outputs:
            a, b = c.density_loss_mult, c.rgb_density_loss_mult * c.density_loss_mult
            ld["density_loss"] = (F.AsymmetricL1.apply(outputs["density2"], outputs["density_thermal"], b, a)
                                  + F.AsymmetricL1.apply(outputs["density"], outputs["density2_thermal"], b, a))
        if c.tv_pixel_loss_mult > 0:
            ld["tv_pixel_loss"] = tv_l
        if c.cross_channel_loss_mult > 0:
            ld["cross_channel_loss"] = cross_l
        return ld

    # ------------------------------------------------------------------------------------------------ fused fast path
    def train_iteration(self, ray_bundle: RayBundle, batch: Dict[str, Tensor], step: int, grad_hook=None, jitters=None, jitters_thermal=None,
                        grad_scaler=None, step_callback: bool = True):
        """Trainer.train_iteration for this model without an autograd tape: callbacks + forward + losses + backward (+ all-reduce) + Adam.
        grad_scaler: optim.DeviceGradScaler -- the reference Trainer's GradScaler semantics (skip on inf / NaN, scale growth / backoff, LR-schedule
        lag; engine/trainer.py:470-495) decided on the device -- or None."""
        cam = ray_bundle.camera_indices.reshape(-1).contiguous()
        return self.engine.train_step(ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous(), cam, batch["image"].to(self.device)[..., :3].contiguous(),
                                      batch["is_thermal"].to(self.device).float().contiguous(), step, jitters, jitters_thermal, grad_hook, grad_scaler=grad_scaler,
                                      step_callback=step_callback)

    # ------------------------------------------------------------------------------------------------ eval
    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """models/base_model.py:177-205: chunks of eval_num_rays_per_chunk rays; outputs reshaped to (H, W, -1)."""
        input_device = camera_ray_bundle.directions.device
        h, w = camera_ray_bundle.origins.shape[:2]
        n = len(camera_ray_bundle)
        chunk = self.config.eval_num_rays_per_chunk
        lists = defaultdict(list)
        for i in range(0, n, chunk):
            rb = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + chunk).to(self.device)
            for k, v in self.forward(rb).items():
                if isinstance(v, Tensor):
                    lists[k].append(v.to(input_device))
        return {k: torch.cat(v).view(h, w, -1) for k, v in lists.items()}

    @torch.no_grad()
    def get_outputs_for_camera(self, camera, obb_box=None) -> Dict[str, Tensor]:
        """models/base_model.py:165-175: one full image of `camera` -- anything with the Cameras fields for ONE camera: camera_to_worlds [3,4],
        fx, fy, cx, cy, width, height, optional distortion_params [6] and camera index (attribute `camera_index`, default 0; it selects the
        pose correction / appearance row, as Cameras.generate_rays(camera_indices=0) + set_camera_indices do for an eval camera)."""
        if obb_box is not None:
            raise NotImplementedError("obb_box cropping is outside the thermal-nerfacto path")
        dev = self.device
        g = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev).reshape(-1)  # noqa: E731
        H, W = int(torch.as_tensor(camera.height).reshape(-1)[0]), int(torch.as_tensor(camera.width).reshape(-1)[0])
        yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        idx = torch.stack([torch.zeros(H * W, dtype=torch.int64, device=dev), yy.reshape(-1), xx.reshape(-1)], dim=1).contiguous()
        dist = getattr(camera, "distortion_params", None)
        dist = None if dist is None else g(dist).reshape(1, 6).contiguous()
        c2w = torch.as_tensor(camera.camera_to_worlds, dtype=torch.float32, device=dev).reshape(1, 3, 4).contiguous()
        o, d, area, norm = ops.raygen(idx, c2w, g(camera.fx)[:1], g(camera.fy)[:1], g(camera.cx)[:1], g(camera.cy)[:1], dist)
        cam_index = int(getattr(camera, "camera_index", 0))
        bundle = RayBundle(origins=o.view(H, W, 3), directions=d.view(H, W, 3), pixel_area=area.view(H, W, 1),
                           camera_indices=torch.full((H, W, 1), cam_index, dtype=torch.int64, device=dev), metadata={"directions_norm": norm.view(H, W, 1)})
        return self.get_outputs_for_camera_ray_bundle(bundle)

    def get_image_metrics_and_images(self, outputs: Dict[str, Tensor], batch: Dict[str, Any]):
        """models/thermal_nerfacto.py:490-564 for one eval image: PSNR / SSIM of the spectrum the image belongs to, and the side-by-side images.
        PSNR = torchmetrics PeakSignalNoiseRatio(data_range=1.0).  SSIM follows the published algorithm with torchmetrics' defaults (11x11
        Gaussian window, sigma 1.5, k1 0.01, k2 0.03, data range from the inputs); torchmetrics is not installed here, so it is NOT pinned
        against it.  LPIPS needs pretrained network weights that are not available offline: the key is omitted.  The accumulation / depth
        images are returned raw (the reference colour-maps them for the viewer: presentation only)."""
        dev = self.device
        is_th = batch["is_thermal"]
        is_thermal_image = bool(is_th) if not hasattr(is_th, "__len__") else bool(torch.as_tensor(is_th).reshape(-1)[0])
        gt = batch["image"].to(dev)[..., :3].float()
        gt_rgbt = rgb_to_rgbt_image(gt.reshape(-1, 3), torch.full((gt.shape[0] * gt.shape[1],), float(is_thermal_image), device=dev)).view(*gt.shape[:2], 4)
        gt_rgb, gt_thermal = gt_rgbt[..., :3], gt_rgbt[..., 3:]
        pred_rgb, pred_th = outputs["rgb"], outputs["rgb_thermal"]
        gt_img = gt_thermal.expand(-1, -1, 3) if is_thermal_image else gt_rgb
        images = {"img": torch.cat([gt_img, pred_rgb, pred_th.expand(-1, -1, 3)], dim=1)}
        if self.config.density_mode == "separate":
            images["accumulation"] = torch.cat([outputs["accumulation"], outputs["accumulation_thermal"]], dim=1)
            images["depth"] = torch.cat([outputs["depth"], outputs["depth_thermal"]], dim=1)
        else:
            images["accumulation"], images["depth"] = outputs["accumulation"], outputs["depth"]
        for i in range(self.config.num_proposal_iterations):
            images[f"prop_depth_{i}"] = outputs[f"prop_depth_{i}"]
        chw = lambda t: torch.moveaxis(t, -1, 0)[None, ...]  # noqa: E731
        metrics: Dict[str, float] = {}
        if not is_thermal_image:
            metrics["psnr_rgb"] = float(_psnr(chw(pred_rgb), chw(gt_rgb)))
            metrics["ssim_rgb"] = float(_ssim(chw(pred_rgb), chw(gt_rgb)))
        else:
            metrics["psnr_thermal"] = float(_psnr(chw(pred_th), chw(gt_thermal)))
            metrics["ssim_thermal"] = float(_ssim(chw(pred_th), chw(gt_thermal)))
        return metrics, images

    def load_model(self, loaded_state: Dict[str, Any]) -> None:
        state = {k.replace("module.", ""): v for k, v in loaded_state["model"].items()}
        self.load_state_dict(state)

    def update_to_step(self, step: int) -> None:
        self.step = step

