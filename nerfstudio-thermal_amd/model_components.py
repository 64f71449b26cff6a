"""Host-side mirrors of the reference's model components for the thermal-nerfacto path: same class names, constructor arguments and
call signatures (model_components/{ray_samplers,renderers,scene_colliders}.py, fields/{density_fields,thermal_nerfacto_field}.py,
cameras/camera_optimizers.py), with every forward routed to the HIP kernels through `ops`.  No CPU fallback anywhere.

These classes serve users who compose the pieces themselves (inference / analysis).  Training goes through ThermalNerfactoModel, which
drives the same kernels as one fused step (engine.RenderEngine) or, under a torch autograd tape, through model._RenderFn.
"""
from __future__ import annotations

from enum import Enum
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .ops import FieldParams, PropNetParams
from .rays import RayBundle, RaySamples, ray_samples_from_level


class FieldHeadNames(Enum):
    """field_components/field_heads.py:28-43 (the two heads this path produces)."""

    RGB = "rgb"
    DENSITY = "density"


class _Node(nn.Module):
    """anonymous container: reproduces the reference's dotted state_dict names (e.g. mlp_base.model.1.layers.0.weight)."""


def register_dotted(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    node.register_parameter(parts[-1], param)


# ------------------------------------------------------------------------------------------------ fields
class HashMLPDensityField(nn.Module):
    """fields/density_fields.py:34-118.  Parameters are views into the model's arena; `encoding.hash_table` and
    `mlp_base.0.hash_table` are the same Parameter, as in the reference."""

    def __init__(self, aabb: Tensor, net: PropNetParams, max_res: int, params: Dict[str, nn.Parameter]):
        super().__init__()
        dev = params["table"].device  # buffers live with the parameters (the reference moves the whole module: DDP broadcasts module states)
        self.register_buffer("aabb", aabb.to(dev))
        self.register_buffer("max_res", torch.tensor(max_res, device=dev))
        self.register_buffer("num_levels", torch.tensor(net.num_levels, device=dev))
        self.register_buffer("log2_hashmap_size", torch.tensor(net.log2_hashmap_size, device=dev))
        self.net = net
        register_dotted(self, "encoding.hash_table", params["table"])
        register_dotted(self, "mlp_base.0.hash_table", params["table"])
        for i, (w, b) in enumerate((("w0", "b0"), ("w1", "b1"))):
            register_dotted(self, f"mlp_base.1.layers.{i}.weight", params[w])
            register_dotted(self, f"mlp_base.1.layers.{i}.bias", params[b])

    def density_from_bins(self, origins: Tensor, directions: Tensor, e_bins: Tensor) -> Tensor:
        return ops.prop_density_fwd(self.net, origins, directions, e_bins)

    def density_fn(self, positions: Tensor, times: Optional[Tensor] = None) -> Tensor:
        """Field.density_fn (fields/base_field.py:48-68): density at explicit positions [...,3] -> [...,1]."""
        flat = positions.reshape(-1, 3).contiguous()
        zeros = torch.zeros((flat.shape[0], 2), device=flat.device)  # start = end = 0 -> position = origin
        d = ops.prop_density_fwd(self.net, flat, torch.zeros_like(flat), zeros)
        return d.reshape(*positions.shape[:-1], 1)

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, None]:
        rb = ray_samples.frustums
        if ray_samples.e_bins is not None:  # samples of whole rays from this package's samplers: positions are formed in-kernel
            d = self.density_from_bins(rb.origins[..., 0, :].contiguous(), rb.directions[..., 0, :].contiguous(), ray_samples.e_bins)
            return d.unsqueeze(-1), None
        return self.density_fn(rb.get_positions()), None

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> dict:
        return {}

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict[FieldHeadNames, Tensor]:
        density, _ = self.get_density(ray_samples)
        return {FieldHeadNames.DENSITY: density}


class ThermalNerfactoField(nn.Module):
    """fields/thermal_nerfacto_field.py:10-99 over fields/nerfacto_field.py:38-348 (density + RGB(T) heads, fused on the device)."""

    def __init__(self, aabb: Tensor, fld: FieldParams, max_res: int, params: Dict[str, nn.Parameter], prefix_names: Dict[str, str],
                 use_average_appearance_embedding: bool = True):
        super().__init__()
        dev = params["table"].device
        self.register_buffer("aabb", aabb.to(dev))
        self.register_buffer("max_res", torch.tensor(max_res, device=dev))
        self.register_buffer("num_levels", torch.tensor(fld.num_levels, device=dev))
        self.register_buffer("log2_hashmap_size", torch.tensor(fld.log2_hashmap_size, device=dev))
        self.fld = fld
        self.use_average_appearance_embedding = use_average_appearance_embedding
        for short, dotted in prefix_names.items():
            register_dotted(self, dotted, params[short])

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict[FieldHeadNames, Tensor]:
        """Field.forward (fields/base_field.py:114-133): {RGB [N,S,C], DENSITY [N,S,1]}."""
        if compute_normals:
            raise NotImplementedError("predict_normals is outside the HIP hot path")
        fr = ray_samples.frustums
        if ray_samples.ndim != 2 or ray_samples.camera_indices is None:
            raise ValueError("ThermalNerfactoField.forward needs a [num_rays, num_samples] RaySamples with camera indices (RayBundle.get_ray_samples / "
                             "the proposal sampler)")
        cam = ray_samples.camera_indices[..., 0, 0].contiguous()
        dens, rgb, _ = ops.field_fwd(self.fld, fr.origins[..., 0, :].contiguous(), fr.directions[..., 0, :].contiguous(), cam, ray_samples.dense_bins(),
                                     training=self.training)
        return {FieldHeadNames.RGB: rgb, FieldHeadNames.DENSITY: dens.unsqueeze(-1)}

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, None]:
        """Density only.  The geometry feature vector the reference returns beside it never leaves the fused kernel."""
        fr = ray_samples.frustums
        d = ops.field_density_fwd(self.fld, fr.origins[..., 0, :].contiguous(), fr.directions[..., 0, :].contiguous(), ray_samples.dense_bins())
        return d.unsqueeze(-1), None

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> Dict[FieldHeadNames, Tensor]:
        out = self.forward(ray_samples)
        return {FieldHeadNames.RGB: out[FieldHeadNames.RGB]}


# ------------------------------------------------------------------------------------------------ collider / camera optimiser
class NearFarCollider(nn.Module):
    """model_components/scene_colliders.py:169-191."""

    def __init__(self, near_plane: float, far_plane: float, reset_near_plane: bool = True, **kwargs) -> None:
        super().__init__()
        self.near_plane, self.far_plane, self.reset_near_plane = near_plane, far_plane, reset_near_plane

    def set_nears_and_fars(self, ray_bundle: RayBundle) -> RayBundle:
        """The two constants as [..., 1] tensors.  They depend on the batch shape only: built once per shape and handed out again (the
        reference builds them per call with ones_like and two multiplications -- three launches in front of every forward); nothing on this
        path writes into a bundle's nears / fars in place."""
        near = self.near_plane if (self.training or not self.reset_near_plane) else 0
        o = ray_bundle.origins
        key = (tuple(o.shape[:-1]), o.device, o.dtype, float(near), float(self.far_plane))
        hit = self.__dict__.get("_nf")
        if hit is None or hit[0] != key:
            ones = torch.ones_like(o[..., 0:1])
            hit = self._nf = (key, ones * near, ones * self.far_plane)
        ray_bundle.nears, ray_bundle.fars = hit[1], hit[2]
        return ray_bundle

    def forward(self, ray_bundle: RayBundle) -> RayBundle:
        return self.set_nears_and_fars(ray_bundle)


class CameraOptimizer(nn.Module):
    """cameras/camera_optimizers.py:89-213, modes 'off' and 'SO3xR3'."""

    def __init__(self, config, num_cameras: int, device, non_trainable_camera_indices: Optional[Tensor] = None, pose_param: Optional[nn.Parameter] = None,
                 **kwargs) -> None:
        super().__init__()
        self.config = config
        self.num_cameras = num_cameras
        self.device = device
        self.suffix = kwargs.get("suffix", "")
        if config.penalty_scale < 0:
            config.mode = "off"
        self.non_trainable_camera_indices = non_trainable_camera_indices
        frozen = torch.zeros(num_cameras, dtype=torch.uint8)
        if non_trainable_camera_indices is not None:
            frozen[non_trainable_camera_indices] = 1
        self.register_buffer("_frozen", frozen.to(device), persistent=False)
        if config.mode != "off":
            assert pose_param is not None
            self.pose_adjustment = pose_param

    def apply_to_raybundle(self, raybundle: RayBundle) -> None:
        if self.config.mode != "off":
            cam = raybundle.camera_indices.reshape(-1).contiguous()
            o, d = ops.pose_apply_fwd(self.pose_adjustment.detach(), self._frozen, cam, raybundle.origins.contiguous(), raybundle.directions.contiguous())
            raybundle.origins, raybundle.directions = o, d

    def get_loss_dict(self, loss_dict: dict) -> None:
        if self.config.mode != "off":
            from .autograd_ops import CameraRegularizer

            loss_dict[f"camera_opt_regularizer{self.suffix}"] = CameraRegularizer.apply(
                self.pose_adjustment, self.config.trans_l2_penalty, self.config.rot_l2_penalty, self.config.penalty_scale)

    def get_metrics_dict(self, metrics_dict: dict) -> None:
        if self.config.mode != "off":
            pa = self.pose_adjustment.detach()  # metrics only: no autograd nodes for two norms per iteration
            metrics_dict[f"camera_opt_translation{self.suffix}"] = pa[:, :3].norm()
            metrics_dict[f"camera_opt_rotation{self.suffix}"] = pa[:, 3:].norm()

    def get_param_groups(self, param_groups: dict, name: str = "camera_opt") -> None:
        if self.config.mode != "off":
            param_groups[name] = [self.pose_adjustment]


# ------------------------------------------------------------------------------------------------ samplers
class UniformLinDispPiecewiseSampler(nn.Module):
    """model_components/ray_samplers.py:225-248 over SpacedSampler :53-128."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__()
        self.num_samples, self.train_stratified, self.single_jitter = num_samples, train_stratified, single_jitter

    def generate_ray_samples(self, ray_bundle: RayBundle, num_samples: Optional[int] = None, jitter: Optional[Tensor] = None) -> RaySamples:
        S = num_samples or self.num_samples
        N = ray_bundle.origins.shape[0]
        if self.train_stratified and self.training and jitter is None:
            if not self.single_jitter:
                raise NotImplementedError("per-sample jitter (single_jitter=False) is outside the HIP hot path")
            jitter = torch.rand(N, device=ray_bundle.origins.device)
        s, e = ops.spaced_bins(ray_bundle.nears, ray_bundle.fars, S, jitter)
        return ray_samples_from_level(ray_bundle, s, e, ray_bundle.nears, ray_bundle.fars)

    forward = generate_ray_samples


class PDFSampler(nn.Module):
    """model_components/ray_samplers.py:251-372 (include_original=False, histogram_padding=0.01)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False, include_original: bool = True,
                 histogram_padding: float = 0.01) -> None:
        super().__init__()
        if include_original:
            raise NotImplementedError("include_original=True is outside the HIP hot path (the proposal sampler passes False)")
        if histogram_padding != 0.01:
            raise NotImplementedError("histogram_padding is fixed at the reference default 0.01 in the kernel")
        self.num_samples, self.train_stratified, self.single_jitter = num_samples, train_stratified, single_jitter

    def generate_ray_samples(self, ray_bundle: RayBundle, ray_samples: RaySamples, weights: Tensor, num_samples: Optional[int] = None,
                             jitter: Optional[Tensor] = None, anneal: float = 1.0) -> RaySamples:
        S = num_samples or self.num_samples
        N = ray_bundle.origins.shape[0]
        if self.train_stratified and self.training and jitter is None:
            jitter = torch.rand(N, device=ray_bundle.origins.device)
        s, e = ops.pdf_resample(ray_samples.dense_spacing_bins(), weights[..., 0].contiguous(), S, anneal, ray_bundle.nears, ray_bundle.fars, jitter)
        return ray_samples_from_level(ray_bundle, s, e, ray_bundle.nears, ray_bundle.fars)

    forward = generate_ray_samples


class ProposalNetworkSampler(nn.Module):
    """model_components/ray_samplers.py:523-618."""

    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32, num_proposal_network_iterations: int = 2,
                 single_jitter: bool = False, update_sched: Callable = lambda x: 1, initial_sampler=None, pdf_sampler=None) -> None:
        super().__init__()
        if num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        self.initial_sampler = initial_sampler or UniformLinDispPiecewiseSampler(single_jitter=single_jitter)
        self.pdf_sampler = pdf_sampler or PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def generate_ray_samples(self, ray_bundle: RayBundle, density_fns: List[Callable]) -> Tuple[RaySamples, List, List]:
        weights_list, ray_samples_list = [], []
        n = self.num_proposal_network_iterations
        weights, ray_samples = None, None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        for i_level in range(n + 1):
            is_prop = i_level < n
            S = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=S)
            else:
                # torch.pow(weights, anneal) is applied inside the resampling kernel
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, weights, num_samples=S, anneal=self._anneal)
            if is_prop:
                fn = density_fns[i_level]
                owner = getattr(fn, "__self__", None)
                if isinstance(owner, HashMLPDensityField):  # fused path: positions are formed in-kernel from origins/directions/bins
                    density = owner.density_from_bins(ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous(), ray_samples.e_bins).unsqueeze(-1)
                else:
                    density = fn(ray_samples.frustums.get_positions())
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        return ray_samples, weights_list, ray_samples_list

    forward = generate_ray_samples


# ------------------------------------------------------------------------------------------------ renderers
class RGBRenderer(nn.Module):
    """model_components/renderers.py:74-245 with background_color='last_sample' (the thermal-nerfacto default)."""

    def __init__(self, background_color="last_sample", num_channels: int = 3) -> None:
        super().__init__()
        if background_color != "last_sample":
            raise NotImplementedError("only background_color='last_sample' is on the HIP hot path")
        self.background_color = background_color
        self.num_channels = num_channels

    def forward(self, rgb: Tensor, weights: Tensor, ray_indices=None, num_rays=None, background_color=None) -> Tensor:
        if ray_indices is not None:
            raise NotImplementedError("packed samples are not produced by the proposal sampler")
        N, S = weights.shape[0], weights.shape[1]
        dummy = torch.zeros((N, S + 1), device=rgb.device)
        return ops.composite_fwd(rgb.contiguous(), weights[..., 0].contiguous(), dummy, self.training, want_depth=False)[0]


class RGBTRenderer(RGBRenderer):
    """model_components/renderers.py:248-425 (4 channels; the thermal background channel is always 0, utils/colors.py:36-48)."""

    def __init__(self, background_color="last_sample") -> None:
        super().__init__(background_color=background_color, num_channels=4)


class AccumulationRenderer(nn.Module):
    """model_components/renderers.py:482-510."""

    def forward(self, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        return torch.sum(weights, dim=-2)


class DepthRenderer(nn.Module):
    """model_components/renderers.py:513-578."""

    def __init__(self, method: str = "median") -> None:
        super().__init__()
        if method not in ("median", "expected"):
            raise NotImplementedError(method)
        self.method = method

    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices=None, num_rays=None) -> Tensor:
        N, S = weights.shape[0], weights.shape[1]
        w = weights[..., 0].contiguous()
        dummy_rgb = torch.zeros((N, S, 1), device=w.device)
        _, _, med, exp = ops.composite_fwd(dummy_rgb, w, ray_samples.dense_bins(), True, want_depth=True)
        return med if self.method == "median" else exp
