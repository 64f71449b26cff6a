"""Orchestration of the hot path over the C ABI: the kernel sequence behind ThermalNerfactoModel.get_outputs
(models/thermal_nerfacto.py:403-489 -> models/nerfacto.py:299-353 -> model_components/ray_samplers.py:577-618) and, for
training, get_loss_dict + backward + Adam as one fused step with no autograd tape.

Everything here is launch plumbing; the arithmetic lives in libthermal_nerf_hip.so.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import ops
from .arena import ParamArena
from .config import ThermalNerfactoModelConfig
from .netparams import field_params, prop_params

# timing experiments: TN_FUSE_SMALL=0 runs the small operators through one entry point per reference seam (tn_weights_fwd, tn_composite_fwd,
# tn_clip_depth, tn_pixel_losses, tn_proposal_losses, tn_pose_apply_fwd/bwd, tn_camera_reg) instead of the fused launches
_FUSE = os.environ.get("TN_FUSE_SMALL", "1") != "0"
# TN_ONE_CALL_BWD=0: the backward of a branch as the individual entry points (render_bwd, weights_bwd, prop_density_bwd x2, field_bwd on torch
# side streams) instead of tn_render_rays_train_bwd (test / A-B aid: the two must agree)
_ONE_CALL_BWD = os.environ.get("TN_ONE_CALL_BWD", "1") != "0"
# TN_TRAIN_STEP_ONE_CALL=0: the fused step through its five library calls instead of tn_train_step (test / A-B aid: the two must agree)
_ONE_CALL_STEP = os.environ.get("TN_TRAIN_STEP_ONE_CALL", "1") != "0"
# TN_NEXT_SAMPLING=0: the one-call step does not run the NEXT iteration's sampling front in its optimiser launch (A/B timing, tests)
_NEXT_SAMPLING = os.environ.get("TN_NEXT_SAMPLING", "1") != "0"


@dataclass
class Level:
    """One sampling level of one branch: what RaySamples carries (cameras/rays.py:251-295) plus the level's density/weights."""

    S: int
    s_bins: Tensor  # [N,S+1] spacing_starts/ends
    e_bins: Tensor  # [N,S+1] frustums.starts/ends
    density: Optional[Tensor] = None  # [N,S]
    weights: Optional[Tensor] = None  # [N,S]
    median: Optional[Tensor] = None  # [N,1]


@dataclass
class Branch:
    """Everything one spectrum branch (rgb / thermal) produces in a forward pass."""

    origins: Tensor
    directions: Tensor
    origins_in: Tensor
    directions_in: Tensor
    levels: List[Level]
    rgb_samples: Tensor  # [N,S,C] per-sample colours
    comp: Tensor  # [N,C]
    accumulation: Tensor
    depth: Tensor
    expected_depth: Tensor
    prop_grad: bool
    fwd_buf: Optional[Tensor] = None  # the one buffer of tn_render_rays_train (fused training forward): what tn_render_rays_train_bwd reads
    prop_enc_saved: bool = False      # ... and it holds the proposal levels' encodings (forward ran with save_prop_enc)


def exp_decay_lr(step: int, lr_init: float, lr_final: float, max_steps: int) -> float:
    """ExponentialDecayScheduler without warm-up (engine/schedulers.py:109-141)."""
    t = float(np.clip(step / max_steps, 0, 1))
    return float(np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


# optimiser table of method_configs["thermal-nerfacto"] (configs/method_configs.py:274-307): group -> (lr, lr_final, max_steps)
OPTIMIZERS = {
    "proposal_networks": (1e-2, 1e-4, 200000),
    "fields": (1e-2, 1e-4, 200000),
    "proposal_networks_thermal": (1e-2, 1e-4, 200000),
    "fields_thermal": (1e-2, 1e-4, 200000),
    "camera_opt": (1e-3, 1e-4, 5000),
    "camera_opt_thermal": (1e-3, 1e-4, 5000),
}


class RenderEngine:
    def __init__(self, cfg: ThermalNerfactoModelConfig, arena: ParamArena, num_images: int, is_thermal_cam: List[int]):
        cfg.validate_for_hip()
        self.cfg, self.arena, self.num_images = cfg, arena, num_images
        dev = arena.device
        self.device = dev
        self.separate = cfg.density_mode == "separate"
        th = torch.tensor([1 if x != 0 else 0 for x in is_thermal_cam], dtype=torch.uint8)
        # camera_optimizer: thermal cameras are non-trainable; camera_optimizer_thermal: RGB cameras are (models/thermal_nerfacto.py:132-144)
        self.frozen_rgb = th.to(dev)
        self.frozen_thermal = (1 - th).to(dev)
        self.props = [prop_params(arena, "proposal_networks", i, cfg, with_grads=True) for i in range(2)]
        self.field = field_params(arena, "field", cfg, with_grads=True)
        self.props_thermal = [prop_params(arena, "proposal_networks_thermal", i, cfg, with_grads=True) for i in range(2)]
        self.field_thermal = field_params(arena, "field_thermal", cfg, with_grads=True) if self.separate else None
        self.pose = arena.view("camera_optimizer.pose_adjustment") if cfg.camera_optimizer.mode != "off" else None
        self.pose_grad = arena.grad_view("camera_optimizer.pose_adjustment") if self.pose is not None else None
        self.pose_thermal = arena.view("camera_optimizer_thermal.pose_adjustment") if cfg.camera_optimizer_thermal.mode != "off" else None
        self.pose_thermal_grad = arena.grad_view("camera_optimizer_thermal.pose_adjustment") if self.pose_thermal is not None else None
        self.counts = list(cfg.num_proposal_samples_per_ray) + [cfg.num_nerf_samples_per_ray]
        # ProposalNetworkSampler state (model_components/ray_samplers.py:564-575)
        self.anneal = 1.0
        self.steps_since_update = 0
        self.sampler_step = 0
        self.adam_step_count = 0
        # The fused step's optimiser launch CONSUMES the gradients (zero behind the read): no arena-wide zero-fill at the start of the next
        # step.  `arena.grads_clean` says the arena's gradient buffer is known to be zero (every writer of the buffer clears it).
        # overlap_adam: the Adam launch over the FIELD groups (470 MB of streaming traffic, ~65 us) runs on a side stream and overlaps the
        # next step's pixel sampling + proposal sampling (which read the proposal networks and the poses only); the field's forward waits for
        # it (tn_render_rays_train's wait event).  OFF: measured 1.09 -> 1.67 ms per step on MI355X / ROCm 7 -- with the second queue active
        # across the step boundary almost every kernel of the step runs 1.5-5x longer (also with the Adam grid capped at 256 blocks; the same
        # two launches on ONE stream cost nothing) -- see profiles/r03_experiments.md.  Kept as an option for runtimes where it pays.
        self.overlap_adam = False
        self._adam_event = None
        # the one-call step runs the NEXT iteration's sampling front (pose correction + proposal sampling of the batch the device data manager has
        # handed over) as co-work of its optimiser launch (TnTrainStep.next_sampling); False: every iteration samples in line
        self.next_sampling = True

    def _side_stream(self, i: int = 0):
        side = self.__dict__.setdefault("_side", {})
        if i not in side:
            side[i] = torch.cuda.Stream(device=self.device)
        return side[i]

    def _uniforms(self) -> "ops.UniformPool":
        """The samplers' per-ray jitter (ray_samplers.py:104-110, 322-330: torch.rand per call), drawn for 32 requests at a time."""
        pool = self.__dict__.get("_rand")
        if pool is None:
            pool = self.__dict__["_rand"] = ops.UniformPool(self.device)
        return pool

    def sync_params(self) -> None:
        """Make the current stream wait for a still-running optimiser launch of the previous train_step (overlap_adam)."""
        ev, self._adam_event = self._adam_event, None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # ---------------------------------------------------------------- sampler schedule
    def update_schedule(self, step: int) -> float:
        c = self.cfg
        # np.clip(np.interp(step, [0, warmup], [0, every]), 1, every) (models/nerfacto.py:283-288) in plain Python: ~15 us of numpy dispatch per call
        w, e = float(c.proposal_warmup), float(c.proposal_update_every)
        x = e if step >= w else (0.0 if step <= 0 else (e / w) * step)  # (slope first, as np.interp evaluates it: bit-identical)
        return min(max(x, 1.0), e)

    def anneal_for_step(self, step: int) -> float:
        """the sampler's histogram-padding exponent at `step` (models/nerfacto.py:271-281)"""
        c = self.cfg
        if not c.use_proposal_weight_anneal:
            return self.anneal
        frac = min(max(step / c.proposal_weights_anneal_max_num_iters, 0.0), 1.0)
        b = c.proposal_weights_anneal_slope
        return b * frac / ((b - 1) * frac + 1)

    def set_anneal_for_step(self, step: int) -> None:
        """set_anneal callback (models/nerfacto.py:271-281)."""
        self.anneal = self.anneal_for_step(step)

    def step_cb(self, step: int) -> None:
        self.sampler_step = step
        self.steps_since_update += 1

    # ---------------------------------------------------------------- forward of one branch
    def render_branch(self, props, fld, pose, frozen, origins: Tensor, directions: Tensor, cam: Tensor, nears: Tensor, fars: Tensor,
                      training: bool, anneal: float, jitters: Optional[List[Tensor]], prop_grad: bool, tag: str = "main", wait_event=None,
                      zero_fill: Optional[Tensor] = None) -> Branch:
        o_in, d_in = origins, directions
        if training and _FUSE:
            # the whole training forward of the branch as ONE call of the C ABI (tn_render_rays_train): the library enqueues the eight
            # launches itself and every result is a view of one allocation -- the host side of a step is what bounds small batches and the
            # drop-in path
            r = ops.render_rays_train(props, fld, pose, frozen, origins, directions, cam, nears, fars, self.counts, anneal, jitters, tag=tag,
                                      wait_event=wait_event, zero_fill=zero_fill,
                                      save_prop_enc=bool(prop_grad) and self.__dict__.get("_bwd_reads_prop_enc", True))
            levels = [Level(S=S, s_bins=lv["s_bins"], e_bins=lv["e_bins"], density=lv["density"], weights=lv["weights"], median=lv["median"])
                      for S, lv in zip(self.counts, r["levels"])]
            return Branch(origins=r["origins"], directions=r["directions"], origins_in=o_in, directions_in=d_in, levels=levels,
                          rgb_samples=r["rgb_samples"], comp=r["rgb"], accumulation=r["accumulation"], depth=r["depth"],
                          expected_depth=r["expected_depth"], prop_grad=prop_grad, fwd_buf=r["buf"], prop_enc_saved=r["prop_enc_saved"])
        first = None
        if training and pose is not None:
            if _FUSE:  # pose correction and the level-0 bins are independent: one launch
                origins, directions, s0, e0 = ops.pose_spaced_bins(pose, frozen, cam, origins, directions, nears, fars, self.counts[0],
                                                                   None if jitters is None else jitters[0])
                first = (s0, e0)
            else:
                origins, directions = ops.pose_apply_fwd(pose, frozen, cam, origins, directions)
        levels: List[Level] = []
        for lvl, S in enumerate(self.counts):
            jit = None if jitters is None else jitters[lvl]
            if lvl == 0:
                s, e = first if first is not None else ops.spaced_bins(nears, fars, S, jit)
            else:
                # weights of the previous level + this level's bins in one launch (get_weights -> PDFSampler, ray_samplers.py:593-611)
                prev = levels[-1]
                prev.weights, prev.median, s, e = ops.weights_resample(prev.e_bins, prev.density, prev.s_bins, S, anneal, nears, fars, jit)
            L = Level(S=S, s_bins=s, e_bins=e)
            if lvl < len(self.counts) - 1:
                L.density = ops.prop_density_fwd(props[lvl], origins, directions, e)
            levels.append(L)
        last = levels[-1]
        last.density, rgb, _ = ops.field_fwd(fld, origins, directions, cam, last.e_bins, training, tag=tag)
        # get_weights + the four renderers (+ the batch-global depth clip) of the last level: one launch
        if _FUSE:
            last.weights, comp, acc, med, exp = ops.render_fwd(last.e_bins, last.density, rgb, training)
        else:
            last.weights, _ = ops.weights_fwd(last.e_bins, last.density)
            comp, acc, med, exp = ops.composite_fwd(rgb, last.weights, last.e_bins, training)
        last.median = med
        return Branch(origins=origins, directions=directions, origins_in=o_in, directions_in=d_in, levels=levels, rgb_samples=rgb, comp=comp,
                      accumulation=acc, depth=med, expected_depth=exp, prop_grad=prop_grad)

    def render_branch_eval(self, props, fld, origins: Tensor, directions: Tensor, cam: Tensor, nears: Tensor, fars: Tensor, anneal: float) -> Branch:
        """render_branch at inference as ONE call of the C ABI (tn_render_rays_eval: the library enqueues the launches back to back)."""
        r = ops.render_rays_eval(props, fld, origins, directions, cam, nears, fars, self.counts, anneal)
        med = [r["prop_depth_0"], r["prop_depth_1"], r["depth"]]
        levels = [Level(S=S, s_bins=lv["s_bins"], e_bins=lv["e_bins"], density=lv["density"], weights=lv["weights"], median=med[i])
                  for i, (S, lv) in enumerate(zip(self.counts, r["levels"]))]
        return Branch(origins=origins, directions=directions, origins_in=origins, directions_in=directions, levels=levels,
                      rgb_samples=r["rgb_samples"], comp=r["rgb"], accumulation=r["accumulation"], depth=r["depth"],
                      expected_depth=r["expected_depth"], prop_grad=False)

    def _nears_fars(self, N: int, training: bool):
        near = self.cfg.near_plane if training else 0.0  # NearFarCollider resets the near plane at inference (scene_colliders.py:186-191)
        key = (N, bool(training))
        cache = self.__dict__.setdefault("_nf_cache", {})
        if key not in cache:  # constants: filled once per batch size, not once per step
            cache.clear()
            cache[key] = (torch.full((N,), near, device=self.device), torch.full((N,), self.cfg.far_plane, device=self.device))
        return cache[key]

    def _zeros_many(self, shapes, fill: bool = True):
        """Zero-initialised tensors carved out of ONE allocation (one fill kernel instead of one per tensor: at 2 ms per step the ~4.5 us
        launch floor of every tiny kernel is visible).  Each view starts on a 256-byte boundary.  fill=False: (views, flat) of an UNINITIALISED
        allocation that somebody else clears (tn_render_rays_train's zero_fill: inside the field's first launch)."""
        sizes = [int(np.prod(s)) for s in shapes]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 63) // 64 * 64
        flat = torch.zeros(tot, device=self.device) if fill else torch.empty(tot, device=self.device)
        views = [flat[o:o + n].view(*s) for o, n, s in zip(offs, sizes, shapes)]
        return views if fill else (views, flat)

    def _accumulator_spec(self, N: int, prop_grads: Dict[str, bool]):
        """keys and shapes of the zero-initialised accumulators of one training iteration (loss_and_backward): the loss vector and its 64 lines,
        then per branch d comp, d weights of the levels that take a gradient, d origins / d directions."""
        zshapes, zkeys = [(16,), (ops.LOSS_LINES, 16)], [("L", ""), ("Lp", "")]
        for sfx in [""] + (["_thermal"] if self.separate else []):
            fld = self.field_thermal if sfx else self.field
            zkeys.append(("d_comp", sfx)); zshapes.append((N, fld.num_channels))
            zkeys.append(("dw2", sfx)); zshapes.append((N, self.counts[2]))
            if prop_grads[sfx]:
                for i in range(2):
                    zkeys.append((f"dw{i}", sfx)); zshapes.append((N, self.counts[i]))
            if (self.pose_thermal if sfx else self.pose) is not None:
                zkeys.append(("d_o", sfx)); zshapes.append((N, 3))
                zkeys.append(("d_d", sfx)); zshapes.append((N, 3))
        if self.separate and self.cfg.density_loss_mult > 0:  # the density loss's four gradient buffers (tn_l1_loss accumulates into them)
            for k in ("g_d2", "g_dt", "g_d", "g_d2t"):
                zkeys.append((k, "")); zshapes.append((N, self.counts[2]))
        return zkeys, zshapes

    @staticmethod
    def _branch_outputs(b: Branch, sfx: str, training: bool) -> Dict[str, object]:
        out = {
            f"rgb{sfx}": b.comp,
            f"accumulation{sfx}": b.accumulation,
            f"depth{sfx}": b.depth,
            f"expected_depth{sfx}": b.expected_depth,
            f"density{sfx}": b.levels[-1].density.unsqueeze(-1),
        }
        for i in range(len(b.levels) - 1):
            out[f"prop_depth_{i}{sfx}"] = b.levels[i].median
        if training:
            out[f"weights_list{sfx}"] = [L.weights.unsqueeze(-1) for L in b.levels]
            out[f"ray_samples_list{sfx}"] = list(b.levels)
        return out

    def get_outputs(self, origins: Tensor, directions: Tensor, cam: Tensor, training: bool, jitters: Optional[List[Tensor]] = None,
                    jitters_thermal: Optional[List[Tensor]] = None, prealloc_accumulators: bool = False):
        """ThermalNerfactoModel.get_outputs.  Returns (outputs dict with the reference's keys, branches for backward).
        prealloc_accumulators (train_step): the iteration's zero-initialised accumulators are allocated here and cleared inside the field's
        first launch of the forward (no fill launch of their own); loss_and_backward picks them up."""
        N = origins.shape[0]
        nears, fars = self._nears_fars(N, training)
        if training and jitters is None:
            jitters = list(self._uniforms().take((3, N)).unbind(0))
        updated = self.steps_since_update > self.update_schedule(self.sampler_step) or self.sampler_step < 10
        wait_ev = None
        if training and _FUSE:
            wait_ev, self._adam_event = self._adam_event, None  # the field's forward (inside tn_render_rays_train) waits for the pending Adam launch
        else:
            self.sync_params()
        zero_fill = None
        self._step_acc = None
        if training and _FUSE and prealloc_accumulators:
            keys, shapes = self._accumulator_spec(N, {"": bool(updated), "_thermal": True})
            views, zero_fill = self._zeros_many(shapes, fill=False)
            self._step_acc = (N, keys, dict(zip(keys, views)))
        if not training and _FUSE:
            b = self.render_branch_eval(self.props, self.field, origins, directions, cam, nears, fars, self.anneal)
        else:
            b = self.render_branch(self.props, self.field, self.pose, self.frozen_rgb, origins, directions, cam, nears, fars, training, self.anneal,
                                   jitters, prop_grad=updated, wait_event=wait_ev, zero_fill=zero_fill)
        self.last_updated = bool(updated)  # did the proposal networks of the RGB sampler get gradients in this forward?
        if updated:  # eval renders included, as ProposalNetworkSampler.generate_ray_samples does (ray_samplers.py:612-613)
            self.steps_since_update = 0
        out = self._branch_outputs(b, "", training)
        branches = {"": b}
        if not self.separate:
            rgbt = out["rgb"]
            out["rgbt"] = rgbt
            out["rgb"] = rgbt[..., :3]
            out["rgb_thermal"] = rgbt[..., 3:]
            return out, branches
        if training and jitters_thermal is None:
            jitters_thermal = list(self._uniforms().take((3, N)).unbind(0))
        # thermal sampler: never receives step_cb -> anneal stays 1.0 and always "updated" (models/thermal_nerfacto.py:222-250)
        if not training and _FUSE:
            bt = self.render_branch_eval(self.props_thermal, self.field_thermal, origins, directions, cam, nears, fars, 1.0)
        else:
            bt = self.render_branch(self.props_thermal, self.field_thermal, self.pose_thermal, self.frozen_thermal, origins, directions, cam, nears,
                                    fars, training, 1.0, jitters_thermal, prop_grad=True)
        out.update(self._branch_outputs(bt, "_thermal", training))
        branches["_thermal"] = bt
        if self.cfg.density_loss_mult > 0 or not training:
            if training:  # density path only, activations kept for the density-only backward
                d2 = ops.field_density_fwd(self.field, bt.origins, bt.directions, bt.levels[-1].e_bins, training=True, tag="cross")
                d2t = ops.field_density_fwd(self.field_thermal, b.origins, b.directions, b.levels[-1].e_bins, training=True, tag="cross")
            else:
                d2 = ops.field_density_fwd(self.field, bt.origins, bt.directions, bt.levels[-1].e_bins)
                d2t = ops.field_density_fwd(self.field_thermal, b.origins, b.directions, b.levels[-1].e_bins)
            out["density2"], out["density2_thermal"] = d2.unsqueeze(-1), d2t.unsqueeze(-1)
        if not training:
            thr = self.cfg.removal_min_density_diff
            last, last_t = b.levels[-1], bt.levels[-1]
            # removal renders (models/thermal_nerfacto.py:460-487): tiny element-wise masks on [N,48] tensors stay in torch,
            # the weights / compositing go through the kernels.  sigma/sigma is NaN where sigma == 0, as in the reference.
            m = ((last.density / last.density - d2t / last.density).abs() < thr).to(torch.float32)
            w, _ = ops.weights_fwd(last.e_bins, (last.density * m).contiguous())
            out["removal"] = ops.composite_fwd(b.rgb_samples, w, last.e_bins, False, want_depth=False)[0]
            m = ((last_t.density / last_t.density - d2 / last_t.density).abs() < thr).to(torch.float32)
            # reference quirk: the thermal removal weights use the RGB branch's ray_samples (models/thermal_nerfacto.py:485)
            w, _ = ops.weights_fwd(last.e_bins, (last_t.density * m).contiguous())
            out["removal_thermal"] = ops.composite_fwd(bt.rgb_samples, w, last.e_bins, False, want_depth=False)[0]
        return out, branches

    # ---------------------------------------------------------------- losses + backward (no autograd tape)
    def loss_and_backward(self, out: Dict[str, object], branches: Dict[str, Branch], cam: Tensor, image: Tensor, is_thermal: Tensor,
                          dp=None, _grads_are_zero: bool = False) -> Dict[str, Tensor]:
        """get_metrics_dict['distortion'] + get_loss_dict (models/thermal_nerfacto.py:253-388) and the gradient of their sum with respect to
        every parameter, accumulated into the arena's gradient buffer.  (_grads_are_zero: train_step's promise, see _set_grad_zero.)"""
        self._set_grad_zero(_grads_are_zero)
        c = self.cfg
        N = image.shape[0]
        dev = self.device
        self.arena.grads_clean = False  # this call accumulates into the arena's gradient buffer
        b = branches[""]
        bt = branches.get("_thermal")
        C = b.comp.shape[1]
        # ---- pixel losses -> d comp
        # every zero-initialised accumulator of the step comes out of one allocation / one fill (see _zeros_many)
        # L: 0 rgb 1 thermal 2 tv 3 cross 4 (scratch) 8 interlevel 9 distortion 10 density 11 camreg 12 camreg_thermal
        # Lp: the loss sums spread over LOSS_LINES 64-byte lines (ops.train_losses), added up into L by ops.losses_finish at the end
        zkeys, zshapes = self._accumulator_spec(N, {sfx: bool(br.prop_grad) for sfx, br in branches.items()} | ({} if self.separate else {"_thermal": False}))
        pre = self.__dict__.get("_step_acc")
        self._step_acc = None
        if pre is not None and pre[0] == N and pre[1] == zkeys:  # allocated by get_outputs(prealloc_accumulators=True), cleared by the forward
            Z = pre[2]
        else:
            Z = dict(zip(zkeys, self._zeros_many(zshapes)))
        L, Lp = Z[("L", "")], Z[("Lp", "")]
        d_comp = Z[("d_comp", "")]
        # the pixel terms ride in the first branch's loss launch (ops.train_losses(pixel=...)): one launch instead of two back to back
        if self.separate:
            d_comp_t = Z[("d_comp", "_thermal")]
            pixel = (b.comp, bt.comp, image, is_thermal, c.thermal_loss_mult, c.tv_pixel_loss_mult, c.cross_channel_loss_mult, d_comp, d_comp_t)
        else:
            pixel = (b.comp[:, :3], b.comp[:, 3:], image, is_thermal, c.thermal_loss_mult, c.tv_pixel_loss_mult, c.cross_channel_loss_mult,
                     d_comp[:, :3], d_comp[:, 3:])
        if not _FUSE:
            ops.pixel_losses(*pixel[:7], L[0:8], *pixel[7:])
            pixel = None
        # ---- proposal losses.  NB (models/thermal_nerfacto.py:363-368): metrics_dict["distortion"] is the SUM over suffixes and is added once per
        # suffix, so in separate mode each branch's distortion enters with 2x distortion_loss_mult.
        nsfx = 2 if self.separate else 1
        grads_w: Dict[str, List[Optional[Tensor]]] = {}
        for sfx, br in branches.items():
            lv = br.levels
            dws: List[Optional[Tensor]] = [None, None, Z[("dw2", sfx)]]
            for i in range(2):
                if br.prop_grad:
                    dws[i] = Z[(f"dw{i}", sfx)]
            # distortion + both interlevel terms (+ the pixel terms): one launch
            props_l = [(lv[i].s_bins, lv[i].weights, dws[i]) for i in range(2)]
            if _FUSE:
                ops.train_losses(lv[2].s_bins, lv[2].weights, props_l, c.distortion_loss_mult * nsfx, c.interlevel_loss_mult, dws[2], Lp, pixel=pixel)
            else:
                ops.proposal_losses(lv[2].s_bins, lv[2].weights, props_l, c.distortion_loss_mult * nsfx, c.interlevel_loss_mult, L[9:10], L[8:9], dws[2])
            pixel = None
            grads_w[sfx] = dws
        # ---- per-branch backward
        # Overlapped data-parallel exchange only where a slice of the arena is final right after its kernel: in separate mode the
        # cross-evaluated densities scatter into the same tables again later, so everything is exchanged by dp.finish() instead.
        pipelined = dp is not None and not self.separate
        d_dens_extra: Dict[str, Optional[Tensor]] = {"": None, "_thermal": None}
        if self.separate and c.density_loss_mult > 0:
            a, bb = c.density_loss_mult, c.rgb_density_loss_mult * c.density_loss_mult
            d2, d2t = out["density2"].squeeze(-1), out["density2_thermal"].squeeze(-1)
            dens, dens_t = b.levels[-1].density, bt.levels[-1].density
            g_d2, g_dt, g_d, g_d2t = Z[("g_d2", "")], Z[("g_dt", "")], Z[("g_d", "")], Z[("g_d2t", "")]  # (zero, from the step's one allocation)
            # a*|d2.detach - dens_t| + b*|d2 - dens_t.detach|  and  a*|dens.detach - d2t| + b*|dens - d2t.detach|   (:336-344)
            ops.l1_loss(d2, dens_t, bb, a, L[10:11], g_d2, g_dt)
            ops.l1_loss(dens, d2t, bb, a, L[10:11], g_d, g_d2t)
            d_dens_extra[""], d_dens_extra["_thermal"] = g_d, g_dt
        for sfx, br in branches.items():
            fld = self.field_thermal if sfx else self.field
            props = self.props_thermal if sfx else self.props
            pose = self.pose_thermal if sfx else self.pose
            pose_grad = self.pose_thermal_grad if sfx else self.pose_grad
            frozen = self.frozen_thermal if sfx else self.frozen_rgb
            lv = br.levels
            want_pos = pose is not None
            d_o = Z[("d_o", sfx)] if want_pos else None
            d_d = Z[("d_d", sfx)] if want_pos else None
            dws = grads_w[sfx]
            dc = d_comp_t if sfx else d_comp
            if _FUSE and _ONE_CALL_BWD and not pipelined and br.fwd_buf is not None and getattr(self, "scatter_events", None) is None:
                # the whole backward of the branch as ONE call of the C ABI (tn_render_rays_train_bwd): renderer backward, field backward with
                # d position and table scatter, both proposal networks on the library's companion streams -- the launches below, enqueued
                # by the library in the same order per stream
                ops.render_rays_train_bwd(props, fld, br.fwd_buf, br.origins, br.directions, cam, self.counts, dc,
                                          [dws[0] if br.prop_grad else None, dws[1] if br.prop_grad else None, dws[2]], d_dens_extra[sfx], d_o, d_d,
                                          tag="main", side_tags=("side0" + sfx, "side1" + sfx), prop_enc_saved=br.prop_enc_saved)
                br._d_o, br._d_d = d_o, d_d
                continue
            if _FUSE:
                d_rgb, d_dens = ops.render_bwd(lv[2].e_bins, lv[2].density, br.rgb_samples, lv[2].weights, dc, dws[2])
            else:
                d_rgb = ops.composite_bwd(br.rgb_samples, lv[2].weights, dc, dws[2])
                d_dens = ops.weights_bwd(lv[2].e_bins, lv[2].density, lv[2].weights, dws[2])
            if d_dens_extra[sfx] is not None:
                d_dens += d_dens_extra[sfx]
            # The proposal networks' backward (own tables, MLPs and scatter; d origins / d directions are accumulated atomically) is
            # independent of the main field's and runs on side streams.
            #   plain step: level 0 on one side stream, level 1 on a second one (both beside the field's backward, which also forks d position
            #   to the library's companion stream) -- measured 1.249 -> 1.216 ms per step against one shared side stream.
            #   data-parallel schedule: BOTH levels, one after the other, on ONE side stream, and nothing else forks (d position runs in line).
            #   The step then keeps three streams busy -- main, this one, RCCL's -- so that with the runtime's DEFAULT four hardware queues every
            #   busy stream has a queue of its own.  Round 4's schedule (side stream + companion stream + level 1 behind the table ranges + two
            #   communicators) ran at 0.89 ms with GPU_MAX_HW_QUEUES=8 and stalled at 1.8-2.1 ms with 4 / 5 / 7: which streams shared a queue
            #   decided a factor of two (profiles/r04_experiments.md; sweep of this schedule: profiles/r05_dp_hwq_sweep.json).
            side = None
            on_side = (0, 1) if pipelined else (0,)
            if br.prop_grad:
                side = self._side_stream()
                main = torch.cuda.current_stream()
                side.wait_stream(main)  # dws[i], d_o, d_d are produced/zeroed on the main stream
                with torch.cuda.stream(side):
                    for i in on_side:
                        dd = ops.weights_bwd(lv[i].e_bins, lv[i].density, lv[i].weights, dws[i])
                        ops.prop_density_bwd(props[i], br.origins, br.directions, lv[i].e_bins, dd, d_o, d_d, tag=f"side{i}")
            if pipelined:
                # main table in level ranges: each range's all-reduce runs beside the scatter of the next one
                ph = ops._lib
                # (d position in line: one stream less -- see above)
                ops.field_bwd_phase(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d, ph.TN_BWD_MLP)
                if _FUSE and pose is not None and not br.prop_grad:
                    # Nothing else adds to d origins / d directions on a step without a proposal update: the pose gradient (+ the loss sums and
                    # the camera regulariser) can be finished NOW, and with it everything behind the table in the arena is final -- MLP
                    # weights, embedding, pose.  Their exchange goes out here, hidden behind the scatter, instead of as one more collective
                    # behind the last table range (~35-50 us between the last fold and the optimiser, whatever the number of ranks).
                    co = c.camera_optimizer_thermal if sfx else c.camera_optimizer
                    ops.pose_bwd_finish(pose, frozen, cam, br.directions_in, d_o, d_d, pose_grad, co.trans_l2_penalty, co.rot_l2_penalty, co.penalty_scale,
                                        L[12:13] if sfx else L[11:12], Lp, L)
                    br._pose_done = True
                    dp.reduce_range(self.arena.layout["field.mlp_base.model.0.hash_table"][0] + self._table_floats(fld), self._camera_hi())
                T2 = 2 * 2**fld.log2_hashmap_size
                t0 = self.arena.layout["field.mlp_base.model.0.hash_table"][0]
                P = N * self.counts[-1]
                # the coarse levels first, exchanged as dense per-cell sums (their table slice is almost all zeros): the longest prefix of levels
                # that are all dense-replica levels at this batch size
                nd = 0
                while getattr(dp, "dense_exchange", False) and nd < fld.num_levels and ops.field_dense_count(fld, P, 0, nd + 1) > 0:
                    nd += 1
                if nd > 0:
                    cells = ops.field_dense_count(fld, P, 0, nd)
                    dense = torch.empty((cells, 2), device=dev)
                    ops.field_bwd_scatter_dense(fld, br.origins, br.directions, lv[2].e_bins, d_o, d_d, 0, nd, dense)
                    dp.reduce_tensor(dense, (t0, t0 + nd * T2), lambda fld=fld, P=P, nd=nd, dense=dense: ops.field_dense_fold(fld, P, 0, nd, dense))
                # one bin pass over every level, then one fold per exchanged level range (with the dense exchange of the coarse levels the
                # remaining levels go range by range through the whole scatter instead: bin + fold per range)
                two_step = nd == 0
                if two_step:
                    ops.field_bwd_phase(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d,
                                        ph.TN_BWD_SCATTER_BIN | ph.TN_BWD_COUNTERS_CLEAN)  # (the MLP phase above left the bucket counters zeroed)
                for lb, le in dp.level_ranges(fld.num_levels - nd):
                    lb, le = lb + nd, le + nd
                    ops.field_bwd_phase(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d,
                                        ph.TN_BWD_SCATTER_FOLD if two_step else ph.TN_BWD_SCATTER, lb, le)
                    dp.reduce_range(t0 + lb * T2, t0 + le * T2)
                ops.field_bwd_phase(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d, ph.TN_BWD_JOIN)
            else:
                # the level-1 network on a second side stream (the package asks the runtime for 8 hardware queues, see __init__.py; with
                # the default 4 a fifth busy stream shares a queue and serialises).  Measured: 1.249 -> 1.216 ms per step.
                side1 = None
                if br.prop_grad:
                    side1 = self._side_stream(1)
                    side1.wait_stream(main)
                    with torch.cuda.stream(side1):
                        dd = ops.weights_bwd(lv[1].e_bins, lv[1].density, lv[1].weights, dws[1])
                        ops.prop_density_bwd(props[1], br.origins, br.directions, lv[1].e_bins, dd, d_o, d_d, tag="side1")
                ev = getattr(self, "scatter_events", None)
                if ev is not None and not sfx:
                    # measurement hook (bench.py: roofline.avg_launch_ms_in_step): the same three phases as one tn_field_bwd call, with a HIP
                    # event pair around the table scatter on the launch stream -- whatever runs beside it in the step still does
                    ph = ops._lib
                    ops.field_bwd_phase(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d, ph.TN_BWD_MLP)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.field_bwd_phase(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d,
                                        ph.TN_BWD_SCATTER | ph.TN_BWD_COUNTERS_CLEAN, 0, fld.num_levels)
                    e1.record()
                    ops.field_bwd_phase(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d, ph.TN_BWD_JOIN)
                    ev.append((e0, e1, bool(br.prop_grad)))
                else:
                    ops.field_bwd(fld, br.origins, br.directions, cam, lv[2].e_bins, d_dens, d_rgb, d_o, d_d)
                if side1 is not None:
                    torch.cuda.current_stream().wait_stream(side1)
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                if pipelined:
                    # both proposal networks' gradients are final: ONE collective for the group, on the same communicator as the table ranges and
                    # issued behind them (a communicator runs its collectives in issue order: ahead of the table ranges it would hold them up until
                    # the whole proposal backward is through, which is why round 4 needed a second communicator)
                    dp.reduce_range(*self.arena.group_range["proposal_networks"])
            br._d_o, br._d_d = d_o, d_d  # cross-evaluation gradients are added below before the pose backward
        if self.separate and c.density_loss_mult > 0:
            # density2 = field at the thermal branch's samples/rays; density2_thermal = field_thermal at the rgb branch's
            # density-only backward: the colour head saw these samples with a zero gradient (models/thermal_nerfacto.py:447-458 calls
            # get_density only), so its backward, its three weight-gradient GEMMs and the embedding rows are skipped
            ops.field_bwd(self.field, bt.origins, bt.directions, cam, bt.levels[-1].e_bins, g_d2, None, bt._d_o, bt._d_d, tag="cross")
            if dp is not None:
                # Data parallel, separate mode: a table is final after its SECOND scatter (own branch + the cross-evaluated density).  The RGB
                # field's 64 MB go out here and travel beside the thermal field's cross backward; the thermal table follows it, the rest at
                # finish().  (DDP's buckets overlap with the backward in every mode: pipelines/base_pipeline.py:281-283.)
                t0 = self.arena.layout["field.mlp_base.model.0.hash_table"][0]
                dp.reduce_range(t0, t0 + self._table_floats(self.field))
            ops.field_bwd(self.field_thermal, b.origins, b.directions, cam, b.levels[-1].e_bins, g_d2t, None, b._d_o, b._d_d, tag="cross")
            if dp is not None:
                t0 = self.arena.layout["field_thermal.mlp_base.model.0.hash_table"][0]
                dp.reduce_range(t0, t0 + self._table_floats(self.field_thermal))
        finished = not _FUSE
        # GradScaler's found_inf from the kernels that write the gradients (see train_step): the last pose launch of the step also scans the
        # small ranges no scatter sees; the flags count as raised only when every camera group's launch went through this path
        kscaler = self.__dict__.get("_kflags_scaler") if (dp is None and _FUSE) else None
        gidx = {g: i for i, g in enumerate(self.arena.optimised_groups)}
        todo = [sfx for sfx, br in branches.items() if (self.pose_thermal if sfx else self.pose) is not None and not getattr(br, "_pose_done", False)]
        handled = set()
        for sfx, br in branches.items():
            pose = self.pose_thermal if sfx else self.pose
            if pose is None:
                continue
            if getattr(br, "_pose_done", False):  # finished early (data-parallel schedule, step without a proposal update)
                finished = True
                continue
            pose_grad = self.pose_thermal_grad if sfx else self.pose_grad
            frozen = self.frozen_thermal if sfx else self.frozen_rgb
            co = c.camera_optimizer_thermal if sfx else c.camera_optimizer
            reg = L[12:13] if sfx else L[11:12]
            if _FUSE:  # pose gradient + regulariser (+ the loss sums, once) in one launch
                extra = {}
                if kscaler is not None and ("camera_opt" + sfx) in gidx:
                    extra = dict(check_grads=self.arena.grads, check_ranges=self._small_grad_ranges() if sfx == todo[-1] else [],
                                 found_inf=kscaler.found_inf, pose_flag=gidx["camera_opt" + sfx])
                    handled.add("camera_opt" + sfx)
                ops.pose_bwd_finish(pose, frozen, cam, br.directions_in, br._d_o, br._d_d, pose_grad, co.trans_l2_penalty, co.rot_l2_penalty,
                                    co.penalty_scale, reg, None if finished else Lp, None if finished else L, **extra)
                finished = True
            else:
                ops.pose_apply_bwd(pose, frozen, cam, br.directions_in, br._d_o, br._d_d, pose_grad)
                ops.camera_reg(pose, co.trans_l2_penalty, co.rot_l2_penalty, co.penalty_scale, reg, pose_grad)
        if not finished:
            ops.losses_finish(Lp, L)
        self._kernel_flags_done = bool(handled) and handled == {g for g in self.arena.optimised_groups if g.startswith("camera_opt")}
        self._last_losses16, self._last_num_rays = L, int(image.shape[0])  # (train_metrics)
        losses = {"rgb_loss": L[0], "thermal_loss": L[1], "tv_pixel_loss": L[2], "cross_channel_loss": L[3], "interlevel_loss": L[8],
                  "distortion_loss": L[9]}
        if self.separate and c.density_loss_mult > 0:
            losses["density_loss"] = L[10]
        if self.pose is not None:
            losses["camera_opt_regularizer"] = L[11]
        if self.separate and self.pose_thermal is not None:
            losses["camera_opt_regularizer_thermal"] = L[12]
        return losses

    def _table_floats(self, fld) -> int:
        return fld.num_levels * 2 * 2**fld.log2_hashmap_size

    def _camera_hi(self) -> int:
        """end of the shared-mode live range behind the main table: field embedding + MLPs, then the camera optimiser's pose"""
        return self.arena.group_range["camera_opt"][1]

    # ---------------------------------------------------------------- GradScaler's found_inf raised by the kernels that write the gradients
    def _small_grad_ranges(self) -> List[Tuple[int, int, int]]:
        """(lo, hi, group index) of everything in the optimised groups that is neither a hash table (its scatter raises the group's flag through
        TnGrid.nonfinite_flag) nor a pose (tn_pose_bwd_finish_check sees its contributions): MLP weights, biases, embeddings -- a few 10^4 floats."""
        hit = self.__dict__.get("_small_ranges")
        if hit is not None:
            return hit
        a = self.arena
        out: List[Tuple[int, int, int]] = []
        for gi, g in enumerate(a.optimised_groups):
            if g.startswith("camera_opt"):
                continue
            cur = None
            for name in sorted(a.group_keys[g], key=lambda n: a.layout[n][0]):
                off, shape = a.layout[name]
                n = int(np.prod(shape))
                if name.endswith("hash_table"):
                    if cur is not None:
                        out.append((cur[0], cur[1], gi)); cur = None
                    continue
                if cur is not None and off - cur[1] < a.ALIGN:  # (alignment padding between two tensors is zero: harmless to scan)
                    cur = (cur[0], off + n)
                else:
                    if cur is not None:
                        out.append((cur[0], cur[1], gi))
                    cur = (off, off + n)
            if cur is not None:
                out.append((cur[0], cur[1], gi))
        self._small_ranges = out
        return out

    def _set_kernel_flags(self, scaler) -> None:
        """Point every grid's nonfinite_flag at its optimiser group's found_inf entry (scaler = None: detach them)."""
        gidx = {g: i for i, g in enumerate(self.arena.optimised_groups)}
        def flag(g):
            return None if (scaler is None or g not in gidx) else scaler.found_inf[gidx[g]:gidx[g] + 1]
        for objs, g in ((self.props, "proposal_networks"), ([self.field], "fields"), (self.props_thermal if self.separate else [], "proposal_networks_thermal"),
                        ([self.field_thermal] if self.field_thermal is not None else [], "fields_thermal")):
            for o in objs:
                if o is not None:
                    o.__dict__["nonfinite_flag"] = flag(g)

    def _set_grad_zero(self, flag: bool) -> None:
        """TnGrid.table_grad_is_zero on every grid of the model: the promise that a table's gradient holds zeros when its scatter starts and is
        scattered into once (the fold then stores instead of adding).  Only train_step can give it -- it owns the whole iteration: the arena's
        gradients are zero when its backward starts, and in shared mode every table sees one scatter.  Everything else (loss_and_backward called
        directly, the autograd nodes of the drop-in path: several backward passes may share one arena) withdraws it."""
        flag = bool(flag) and not self.separate
        if self.__dict__.get("_grad_zero_flag", False) == flag:
            return
        self._grad_zero_flag = flag
        for o in list(self.props) + [self.field] + (list(self.props_thermal) if self.separate else []) + (
                [self.field_thermal] if self.field_thermal is not None else []):
            if o is not None:
                o.__dict__["grad_is_zero"] = flag

    # ---------------------------------------------------------------- optimiser
    def optimizer_step(self, lr_overrides: Optional[Dict[str, float]] = None, scheduled: bool = True, skip_groups=(), ranges=None,
                       grad_scaler=None, skipped_have_no_grads: bool = False, flag_reduce=None) -> None:
        """One Adam step per optimiser group over its contiguous arena range (engine/optimizers.py; configs/method_configs.py:274-307).

        torch.optim.Adam skips parameters whose .grad is None and advances its per-parameter step count (bias correction) only when it
        updates them.  The only parameters that have no gradient in some iterations are the proposal networks on steps where the sampler
        ran them under no_grad (ray_samplers.py:605-610): `skip_groups` names them, and each group keeps its own Adam step count.  The LR
        schedule (LambdaLR) advances every iteration for every group.

        grad_scaler (optim.DeviceGradScaler): GradScaler semantics on the device -- the step is a no-op when the gradients hold an inf / NaN,
        bias corrections and LR schedule are evaluated at (count - skipped steps), the scale is updated (engine/trainer.py:470-495)."""
        self.adam_step_count += 1
        if not hasattr(self, "group_steps"):
            self.group_steps = {}
        a = self.arena
        hyper = {}
        for gname in a.optimised_groups:
            if gname in skip_groups:
                continue
            self.group_steps[gname] = self.group_steps.get(gname, 0) + 1
            lr0, lr_final, max_steps = OPTIMIZERS[gname]
            # LambdaLR: the lr used at iteration k (1-based) is the schedule evaluated at k-1
            lr = exp_decay_lr(self.adam_step_count - 1, lr0, lr_final, max_steps) if scheduled else lr0
            if lr_overrides and gname in lr_overrides:
                lr = lr_overrides[gname]
            hyper[gname] = (self.group_steps[gname], lr)
        if ranges is None:
            # One launch for every stepped group (two with overlap_adam), through the entry point that keeps every decision on the device:
            #   * the launch CONSUMES the gradients (zero behind the read): the next step starts without a zero-fill of the arena;
            #   * with a grad scaler: per-group non-finite check, skip / bias-correction / LR-schedule lag on the device, scale update.
            scaler = grad_scaler if (grad_scaler is not None and grad_scaler.enabled) else None
            names = list(hyper)
            gidx = {g: i for i, g in enumerate(a.optimised_groups)}
            # GradScaler decides per optimiser = per parameter group: one flag per group.  When the kernels that WRITE the gradients have raised the
            # flags already (table scatters through TnGrid.nonfinite_flag, everything else in tn_pose_bwd_finish_check) no pass over the arena is
            # needed; otherwise all groups are checked in one launch.
            kernel_flags = scaler is not None and self.__dict__.pop("_kernel_flags_done", False)
            if scaler is not None and not kernel_flags:
                scaler.check_ranges(a.grads, [a.group_range[g] for g in names], [gidx[g] for g in names])
            on_device_lr = scaler is not None and scheduled and not lr_overrides

            def launch(groups, fused_update=False):
                if not groups:
                    return
                if on_device_lr:  # lr_init + (lr_final, max_steps), evaluated at count - lag on the device
                    rng = [a.group_range[g] + (hyper[g][0], OPTIMIZERS[g][0]) for g in groups]
                    sched = [(OPTIMIZERS[g][1], OPTIMIZERS[g][2]) for g in groups]
                else:
                    rng, sched = [a.group_range[g] + hyper[g] for g in groups], None
                ops.adam_step_ranges_amp(a.params, a.grads, a.exp_avg, a.exp_avg_sq, rng, eps=1e-15,
                                         found_inf=scaler.found_inf if scaler is not None else None, flags=[gidx[g] for g in groups] if scaler is not None else None,
                                         skipped=scaler.skipped if scaler is not None else None, lag_index=scaler.lag_index if scaler is not None else -1,
                                         count_skip=scaler is not None, schedule=sched, sched_step=self.adam_step_count - 1, zero_grads=True,
                                         scaler_update=scaler.fused_update_args() if fused_update else None)

            # the stepped groups' gradients are consumed.  A skipped group keeps whatever it holds: the buffer only counts as clean when the caller
            # vouches that the skipped groups received no gradient (train_step: proposal networks on an iteration the sampler ran them under no_grad)
            self.arena.grads_clean = (not skip_groups) or skipped_have_no_grads
            if self.overlap_adam:
                big = [g for g in names if g.startswith("fields")]
                launch([g for g in names if not g.startswith("fields")])
                main, side = torch.cuda.current_stream(), self._side_stream(5)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    launch(big)
                    if scaler is not None:
                        scaler.update()  # behind BOTH launches (it clears the flags they read): the side stream waited for the first
                    ev = torch.cuda.Event()
                    ev.record(side)
                self._adam_event = ev
            else:
                # GradScaler.update() rides on the Adam launch (its last block to finish performs it): one launch less at the end of the step
                fuse = scaler is not None and bool(names)
                launch(names, fused_update=fuse)
                if scaler is not None and not fuse:
                    scaler.update()
            return
        if grad_scaler is not None and grad_scaler.enabled:
            # GradScaler semantics with per-range pieces (the sharded-optimiser schedule: this rank's pieces of the reduce-scattered slices + the
            # replicated small ones).  GradScaler decides per optimiser over the WHOLE gradient, and a non-finite value in a reduce-scattered
            # slice reaches its owner only: every rank checks what it is about to apply, the flags are MAX-reduced over the ranks (flag_reduce:
            # ShardedGradReducer.reduce_flags, one collective of num_groups floats), then ONE Adam launch over all pieces decides on the device.
            gidx = {g: i for i, g in enumerate(a.optimised_groups)}
            pieces = []
            for lo, hi in ranges:  # (the generator waits for every exchange on the stream as it yields)
                gname = next(g for g in a.optimised_groups if a.group_range[g][0] <= lo and hi <= a.group_range[g][1])
                if gname in hyper and hi > lo:
                    pieces.append((lo, hi, gname))
            if len(pieces) > 8:
                raise NotImplementedError("GradScaler semantics for more than 8 Adam pieces per iteration (coarser level ranges keep it below)")
            if pieces:
                grad_scaler.check_ranges(a.grads, [(lo, hi) for lo, hi, _ in pieces], [gidx[g] for _, _, g in pieces])
            if flag_reduce is not None:
                flag_reduce(grad_scaler.found_inf)
            if pieces:
                on_device_lr = scheduled and not lr_overrides
                if on_device_lr:
                    rng = [(lo, hi, hyper[g][0], OPTIMIZERS[g][0]) for lo, hi, g in pieces]
                    sched = [(OPTIMIZERS[g][1], OPTIMIZERS[g][2]) for _, _, g in pieces]
                else:
                    rng, sched = [(lo, hi) + hyper[g] for lo, hi, g in pieces], None
                ops.adam_step_ranges_amp(a.params, a.grads, a.exp_avg, a.exp_avg_sq, rng, eps=1e-15, found_inf=grad_scaler.found_inf,
                                         flags=[gidx[g] for _, _, g in pieces], skipped=grad_scaler.skipped, lag_index=grad_scaler.lag_index, count_skip=True,
                                         schedule=sched, sched_step=self.adam_step_count - 1, zero_grads=False, scaler_update=grad_scaler.fused_update_args())
            else:
                grad_scaler.update()
            return
        for lo, hi in ranges:  # each range lies inside one optimiser group (Adam is element-wise: any partition of a group is the same update)
            gname = next(g for g in a.optimised_groups if a.group_range[g][0] <= lo and hi <= a.group_range[g][1])
            if gname not in hyper:
                continue
            step, lr = hyper[gname]
            ops.adam_step(a.params[lo:hi], a.grads[lo:hi], a.exp_avg[lo:hi], a.exp_avg_sq[lo:hi], step, lr, eps=1e-15)

    # ---------------------------------------------------------------- checkpoint of the fused path's optimiser state
    def optimizer_state_dict(self, grad_scaler=None) -> Dict[str, object]:
        """What Trainer.save_checkpoint stores beside the model (engine/trainer.py:424-447: "optimizers" = torch.optim.Adam.state_dict()
        per parameter group, "schedulers" = LambdaLR.state_dict()), read out of the arena: state[i] = {step, exp_avg, exp_avg_sq} for the i-th
        parameter of the group in get_param_groups() order.  Plus the sampler's update counters, which the reference loses on resume."""
        self.sync_params()
        a = self.arena
        steps = getattr(self, "group_steps", {})
        opt, sched = {}, {}
        # With a grad scaler the host counters include the iterations the DEVICE skipped (inf / NaN gradients) and the schedule lag; what the
        # reference Trainer would have written for the same history excludes them (torch's Adam `step` does not advance on a skipped step, the
        # schedulers are not stepped when the scale dropped: engine/trainer.py:488-495) -- and so do the bias corrections / LR the next launch uses.
        sc = grad_scaler.state_dict() if (grad_scaler is not None and getattr(grad_scaler, "enabled", True)) else None
        lag = int(sc["schedule_lag"]) if sc else 0
        epoch = max(self.adam_step_count - lag, 0)
        for gi, g in enumerate(a.optimised_groups):
            lr0, lr_final, max_steps = OPTIMIZERS[g]
            k = int(steps.get(g, 0)) - (int(sc["skipped"][gi]) if sc else 0)
            state = {}
            if k > 0:
                for i, name in enumerate(a.group_keys[g]):
                    state[i] = {"step": torch.tensor(float(k)), "exp_avg": a._view(a.exp_avg, name).detach().clone(),
                                "exp_avg_sq": a._view(a.exp_avg_sq, name).detach().clone()}
            opt[g] = {"state": state, "param_groups": [{"lr": exp_decay_lr(epoch, lr0, lr_final, max_steps), "betas": (0.9, 0.999),
                                                        "eps": 1e-15, "weight_decay": 0, "amsgrad": False, "initial_lr": lr0,
                                                        "params": list(range(len(a.group_keys[g])))}]}
            sched[g] = {"last_epoch": epoch, "_step_count": epoch + 1, "base_lrs": [lr0]}
        out = {"optimizers": opt, "schedulers": sched,
               "sampler": {"steps_since_update": self.steps_since_update, "step": self.sampler_step, "anneal": self.anneal}}
        if sc:
            # the counters themselves are folded into step / last_epoch above: a resumed scaler starts them at zero
            out["scalers"] = {**sc, "skipped": [0] * len(sc["skipped"]), "schedule_lag": 0}
        return out

    def load_optimizer_state_dict(self, state: Dict[str, object], grad_scaler=None) -> None:
        """Inverse of optimizer_state_dict; also accepts a checkpoint written by the reference Trainer (same layout, no "sampler" entry; its
        "scalers" entry is torch.amp.GradScaler's state_dict, which DeviceGradScaler.load_state_dict takes as it is)."""
        if grad_scaler is not None and state.get("scalers"):
            grad_scaler.load_state_dict(state["scalers"])
        a = self.arena
        self.group_steps = {}
        for g, od in state.get("optimizers", {}).items():
            if g not in a.group_keys:
                continue
            for i, st in od.get("state", {}).items():
                name = a.group_keys[g][int(i)]
                a._view(a.exp_avg, name).copy_(torch.as_tensor(st["exp_avg"]).to(a.device, torch.float32).reshape(a.layout[name][1]))
                a._view(a.exp_avg_sq, name).copy_(torch.as_tensor(st["exp_avg_sq"]).to(a.device, torch.float32).reshape(a.layout[name][1]))
                self.group_steps[g] = max(self.group_steps.get(g, 0), int(float(st["step"])))
        sched = state.get("schedulers", {})
        if sched:
            self.adam_step_count = int(max(int(v.get("last_epoch", 0)) for v in sched.values()))
        else:
            self.adam_step_count = max(self.group_steps.values(), default=0)
        smp = state.get("sampler")
        if smp:
            self.steps_since_update, self.sampler_step, self.anneal = int(smp["steps_since_update"]), int(smp["step"]), float(smp["anneal"])

    def _train_step_one_call(self, origins: Tensor, directions: Tensor, cam: Tensor, image: Tensor, is_thermal: Tensor, jitters, scaler,
                             step: Optional[int] = None) -> Dict[str, Tensor]:
        """The iteration of train_step (shared density, camera optimiser, device-side GradScaler, no data-parallel exchange) as ONE library call,
        tn_train_step: what get_outputs + loss_and_backward + optimizer_step enqueue through five calls, with their bookkeeping done here.
        The five-call path stays the reference (TN_TRAIN_STEP_ONE_CALL=0; tests/test_trainer_sequence_gpu.py compares the two)."""
        a, c = self.arena, self.cfg
        N = origins.shape[0]
        call = self.__dict__.get("_step_call")
        if call is None or call.scaler is not scaler or call.arena_tensors[1] is not a.grads:
            co = c.camera_optimizer
            gidx = {g: i for i, g in enumerate(a.optimised_groups)}
            call = self._step_call = ops.TrainStepCall(
                self.props, self.field, self.pose, self.frozen_rgb, self.pose_grad, (co.trans_l2_penalty, co.rot_l2_penalty, co.penalty_scale), self.counts,
                (c.thermal_loss_mult, c.tv_pixel_loss_mult, c.cross_channel_loss_mult, c.distortion_loss_mult, c.interlevel_loss_mult),
                (a.params, a.grads, a.exp_avg, a.exp_avg_sq), self._small_grad_ranges(), gidx["camera_opt"], scaler)
            self._step_gidx = gidx
        gidx = self._step_gidx
        nears, fars = self._nears_fars(N, True)
        updated = self.steps_since_update > self.update_schedule(self.sampler_step) or self.sampler_step < 10
        # The previous call may have run THIS batch's sampling front (pose correction + both proposal levels) in its optimiser launch
        # (TnTrainStep.next_sampling): valid when the batch is the one it was planned for, the sampler's state is what was predicted, and nobody has
        # written the parameters through torch since (our kernels do not move the version counter; copy_ / load_state_dict do).
        plan, fwd_buf = self.__dict__.pop("_planned", None), None
        if plan is not None and jitters is None and plan["call"] is call and plan["N"] == N and plan["step"] == step \
                and plan["ptrs"] == (origins.data_ptr(), directions.data_ptr(), cam.data_ptr()) and plan["anneal"] == float(self.anneal) \
                and plan["updated"] == bool(updated) and plan["version"] == a.params._version:
            jitters, fwd_buf = plan["jitters"], plan["buf"]
        if jitters is None:
            jitters = list(self._uniforms().take((3, N)).unbind(0))
            drew = True
        else:
            drew = fwd_buf is not None
        # ... and this call plans the next one: the batch the data manager has handed over (ops.sample_rays_deferred), the jitter of the next iteration
        # (drawn now: the same sequence of draws, one iteration early), its anneal exponent and whether its proposal networks take a gradient
        next_plan = None
        pend = ops._PENDING_SAMPLE
        if _NEXT_SAMPLING and self.next_sampling and drew and step is not None and pend is not None and int(pend[0].num_rays) == N and N % 4 == 0:
            since = (0 if updated else self.steps_since_update) + 1  # (what step_cb leaves behind this iteration)
            n_updated = since > self.update_schedule(step) or step < 10
            next_plan = (list(self._uniforms().take((3, N)).unbind(0)), self.anneal_for_step(step + 1), bool(n_updated))
        keys, shapes = self._accumulator_spec(N, {"": bool(updated), "_thermal": False})
        views, flat = self._zeros_many(shapes, fill=False)  # cleared inside the field's first launch
        acc = {k[0]: v for k, v in zip(keys, views)}
        # optimiser bookkeeping of optimizer_step: Adam step counts per group, LR schedule position (evaluated on the device at count - lag).
        # Worked out on copies and COMMITTED only after the library accepted the call: tn_train_step validates every stage's arguments and
        # workspace sizes before its first launch, so a refused call (TN_EINVAL) has enqueued nothing and must leave the host counters where the
        # device's are.
        if not hasattr(self, "group_steps"):
            self.group_steps = {}
        group_steps = dict(self.group_steps)
        ranges = []
        for g in a.optimised_groups:
            if g == "proposal_networks" and not updated:
                continue  # ran under no_grad this iteration (ray_samplers.py:605-610): not stepped, keeps its step count
            group_steps[g] = group_steps.get(g, 0) + 1
            lr0, lr_final, max_steps = OPTIMIZERS[g]
            lo, hi = a.group_range[g]
            ranges.append((lo, hi, group_steps[g], lr0, lr_final, max_steps, gidx[g]))
        self._set_grad_zero(True)  # (train_step made sure the arena's gradients are zero; shared mode: one scatter per table)
        # (the library refuses what it can before its first launch; a refusal or launch error further in leaves gradients behind that no optimiser
        # launch consumed: the buffer only counts as clean again once the call has returned)
        a.grads_clean = False
        try:
            call.run(origins, directions, cam, image, is_thermal, nears, fars, self.anneal, jitters, bool(updated), flat, acc, ranges, self.adam_step_count,
                     fwd_buf=fwd_buf, next_plan=next_plan)
        finally:
            self._set_grad_zero(False)  # the promise holds for this call's scatters only -- also when the call was refused
        if next_plan is not None and call.next_buf is not None:
            o2, d2, c2 = pend[1][3], pend[1][4], pend[1][5]  # (sample_rays_deferred: (u, cameras, cache) + (origins, directions, camera_indices, ...))
            self._planned = {"call": call, "N": N, "step": step + 1, "ptrs": (o2.data_ptr(), d2.data_ptr(), c2.data_ptr()), "anneal": float(next_plan[1]),
                             "updated": next_plan[2], "jitters": next_plan[0], "buf": call.next_buf, "version": a.params._version}
        self.adam_step_count += 1
        self.group_steps = group_steps
        a.grads_clean = True  # the Adam launch consumed the gradients of every group that received any
        self.last_updated = bool(updated)
        if updated:
            self.steps_since_update = 0
        L = acc["L"]
        self._last_losses16, self._last_num_rays = L, int(N)  # (train_metrics)
        return {"rgb_loss": L[0], "thermal_loss": L[1], "tv_pixel_loss": L[2], "cross_channel_loss": L[3], "interlevel_loss": L[8],
                "distortion_loss": L[9], "camera_opt_regularizer": L[11]}

    def _metric_poses(self):
        return [p for p in ((self.pose, self.pose_thermal) if self.separate else (self.pose,)) if p is not None]

    def pose_metrics(self) -> Dict[str, Tensor]:
        """camera_opt_translation / camera_opt_rotation (cameras/camera_optimizers.py:197-202) of the CURRENT pose corrections, one small launch.
        The reference's metrics_dict holds them as they are when the forward runs -- before the iteration's optimiser step: take them before
        train_step and hand them to train_metrics."""
        poses = self._metric_poses()
        out: Dict[str, Tensor] = {}
        if poses:
            dummy = self.__dict__.get("_metric_ones")
            if dummy is None:
                dummy = self._metric_ones = torch.ones(16, device=poses[0].device)
            m = torch.empty(8, device=poses[0].device)
            ops.train_metrics(dummy, 4, 1.0, poses, m)
            for k, sfx in enumerate(("", "_thermal")[:len(poses)]):
                out[f"camera_opt_translation{sfx}"], out[f"camera_opt_rotation{sfx}"] = m[2 + 2 * k], m[3 + 2 * k]
        return out

    def train_metrics(self, pose_metrics: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        """metrics_dict of the iteration train_step just enqueued (models/thermal_nerfacto.py:253-282): the PSNR per spectrum from the pixel-loss
        sums and the distortion metric -- one small launch (tn_train_metrics) behind the step, no host synchronisation -- plus `pose_metrics`
        (see there).  Call before the next train_step (the loss vector is the step's accumulator)."""
        L = self.__dict__.get("_last_losses16")
        if L is None:
            raise RuntimeError("train_metrics: no train_step has run")
        c = self.cfg
        m = torch.empty(8, device=L.device)
        ops.train_metrics(L, self._last_num_rays, c.thermal_loss_mult, [], m)
        out = {"psnr_rgb": m[0], "psnr_thermal": m[1]}
        nsfx = 2 if self.separate else 1
        if c.distortion_loss_mult > 0:
            out["distortion"] = L[9] / (c.distortion_loss_mult * nsfx)
        out.update(pose_metrics or {})
        return out

    def train_step(self, origins: Tensor, directions: Tensor, cam: Tensor, image: Tensor, is_thermal: Tensor, step: int,
                   jitters=None, jitters_thermal=None, grad_hook=None, scheduled: bool = True, grad_scaler=None,
                   step_callback: bool = True) -> Dict[str, Tensor]:
        """Trainer.train_iteration (engine/trainer.py:455-499) for this model: callbacks, forward, losses, backward, (all-reduce), Adam.
        grad_scaler: optim.DeviceGradScaler or None (see optimizer_step).
        step_callback=False: the caller runs the model's AFTER_TRAIN_ITERATION callback itself (the reference Trainer's loop does,
        engine/trainer.py:258-276: trainer.FusedTrainerMixin) -- the sampler's update counter must advance once per iteration."""
        self.set_anneal_for_step(step)
        if grad_scaler is not None:
            grad_scaler.begin_step()
        # found_inf by the kernels that write the gradients -- only when the gradients are final where they are written (no data-parallel exchange
        # behind the backward: with one, an inf on another rank arrives through the all-reduce and the explicit check after it stays)
        kflags = grad_scaler if (grad_scaler is not None and grad_scaler.enabled and grad_hook is None and _FUSE) else None
        if self.__dict__.get("_kflags_scaler") is not kflags:
            self._set_kernel_flags(kflags)
            self._kflags_scaler = kflags
        if not self.arena.grads_clean:  # (the previous step's optimiser launch consumed the gradients: nothing to fill)
            self.sync_params()
            self.arena.zero_grad()
        if (_ONE_CALL_STEP and _FUSE and _ONE_CALL_BWD and kflags is not None and scheduled and not self.separate and self.pose is not None
                and not self.overlap_adam and getattr(self, "scatter_events", None) is None
                and self.field.num_channels == 4 and "camera_opt" in self.arena.optimised_groups):
            losses = self._train_step_one_call(origins, directions, cam, image, is_thermal, jitters, grad_scaler, step)
            self._set_grad_zero(False)  # the promise holds for this iteration's scatters only (anybody may call the ops on these grids next)
            if step_callback:
                self.step_cb(step)
            return losses
        # (the overlapped data-parallel schedule runs the proposal networks' backward through the per-network entry points, which gather again:
        # the forward then need not keep their encodings)
        self._bwd_reads_prop_enc = not (grad_hook is not None and getattr(grad_hook, "pipelined", False) and not self.separate) and _ONE_CALL_BWD
        try:
            out, branches = self.get_outputs(origins, directions, cam, True, jitters, jitters_thermal, prealloc_accumulators=True)
        finally:
            self._bwd_reads_prop_enc = True
        if grad_hook is not None and getattr(grad_hook, "pipelined", False):
            # data-parallel gradient all-reduce overlapped with the backward pass (parallel.OverlappedGradReducer)
            grad_hook.begin(self.arena)
            losses = self.loss_and_backward(out, branches, cam, image, is_thermal, dp=grad_hook, _grads_are_zero=True)
            # proposal networks that got no gradient this step are not stepped either: nothing to exchange for them
            skip = () if branches[""].prop_grad else ("proposal_networks",)
            idle = [self.arena.group_range[g] for g in skip]
            if getattr(grad_hook, "adam_per_range", False):
                # Adam range by range, each as soon as its exchange has landed: the update of the first table levels runs while the last
                # ones are still on the wire (one more launch per range on the host)
                self.optimizer_step(scheduled=scheduled, skip_groups=skip, ranges=grad_hook.finish_iter(skip=idle), grad_scaler=grad_scaler,
                                    flag_reduce=getattr(grad_hook, "reduce_flags", None))
                if getattr(grad_hook, "sharded", False):
                    # optimiser state sharded over the ranks (parallel.ShardedGradReducer): the ranges above were this rank's pieces; now the
                    # updated parameters of every sharded slice are all-gathered
                    grad_hook.gather_params()
            else:
                grad_hook.finish(skip=idle)
                self.optimizer_step(scheduled=scheduled, skip_groups=skip, grad_scaler=grad_scaler, skipped_have_no_grads=True)
        else:
            losses = self.loss_and_backward(out, branches, cam, image, is_thermal, _grads_are_zero=True)
            if grad_hook is not None:
                grad_hook(self.arena)  # data-parallel gradient all-reduce, after the backward pass
            skip = () if branches[""].prop_grad else ("proposal_networks",)
            self.optimizer_step(scheduled=scheduled, skip_groups=skip, grad_scaler=grad_scaler, skipped_have_no_grads=True)
        self._set_grad_zero(False)  # the promise holds for this iteration's scatters only
        if step_callback:
            self.step_cb(step)
        return losses
