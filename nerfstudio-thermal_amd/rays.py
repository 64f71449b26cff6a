"""Boundary types of the hot path: RayBundle / RaySamples / Frustums with the reference's field names, batch semantics and methods
(cameras/rays.py:32-295 over utils/tensor_dataclass.py:27-350, restated in tensor_dataclass.py): broadcast on construction, `[...]`
indexing, reshape / flatten / broadcast_to / to over the batch dimensions of every field, dict fields and the nested Frustums included.
`RaySamples.get_weights` runs on the device through the C ABI.  The kernels take a level's samples as dense per-ray bins [N, S+1]; a
RaySamples produced by this package carries them in the plain attributes `s_bins` / `e_bins` (not dataclass fields: they do not follow
indexing or reshaping, and are rebuilt from the frustums when missing).
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from . import ops
from .tensor_dataclass import TensorDataclass


@dataclass
class Frustums(TensorDataclass):
    origins: Tensor  # [*bs, 3]
    directions: Tensor  # [*bs, 3]
    starts: Tensor  # [*bs, 1]
    ends: Tensor  # [*bs, 1]
    pixel_area: Tensor  # [*bs, 1]
    offsets: Optional[Tensor] = None  # [*bs, 3]

    def get_positions(self) -> Tensor:
        """cameras/rays.py:49-58 (element-wise; evaluated by torch on the device: the fused path recomputes positions in-kernel)."""
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        if self.offsets is not None:
            pos = pos + self.offsets
        return pos

    def get_start_positions(self) -> Tensor:
        return self.origins + self.directions * self.starts

    def set_offsets(self, offsets) -> None:
        self.offsets = offsets


@dataclass
class RaySamples(TensorDataclass):
    frustums: Frustums
    camera_indices: Optional[Tensor] = None  # [*bs, 1]
    deltas: Optional[Tensor] = None  # [*bs, 1]
    spacing_starts: Optional[Tensor] = None  # [*bs, num_samples, 1]
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None

    # dense per-ray bins of the level for the kernels (plain attributes, see the module docstring)
    s_bins = None
    e_bins = None

    def dense_bins(self) -> Tensor:
        """[N, S+1] euclidean bin edges of a [N, S] batch of contiguous frustums (end of sample i = start of sample i+1)."""
        if self.e_bins is None:
            if self.ndim != 2:
                raise ValueError("the kernels take ray samples as a [num_rays, num_samples] batch")
            self.e_bins = torch.cat([self.frustums.starts[..., 0], self.frustums.ends[..., -1:, 0]], dim=-1).contiguous()
        return self.e_bins

    def dense_spacing_bins(self) -> Tensor:
        """[N, S+1] bin edges in normalised (spacing) coordinates, from spacing_starts / spacing_ends."""
        if self.s_bins is None:
            if self.ndim != 2 or self.spacing_starts is None or self.spacing_ends is None:
                raise ValueError("PDF resampling needs [num_rays, num_samples] ray samples that carry spacing_starts / spacing_ends")
            self.s_bins = torch.cat([self.spacing_starts[..., 0], self.spacing_ends[..., -1:, 0]], dim=-1).contiguous()
        return self.s_bins

    def get_weights(self, densities: Tensor) -> Tensor:
        """RaySamples.get_weights (cameras/rays.py:128-150) -> tn_weights_fwd.  densities [N,S,1] -> weights [N,S,1]."""
        w, _ = ops.weights_fwd(self.dense_bins(), densities[..., 0].contiguous())
        return w.unsqueeze(-1)


@dataclass
class RayBundle(TensorDataclass):
    origins: Tensor  # [*batch, 3]
    directions: Tensor  # [*batch, 3]
    pixel_area: Tensor  # [*batch, 1]
    camera_indices: Optional[Tensor] = None  # [*batch, 1]
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    def set_camera_indices(self, camera_index: int) -> None:
        self.camera_indices = torch.ones_like(self.origins[..., 0:1]).long() * camera_index

    def __len__(self) -> int:
        return torch.numel(self.origins) // self.origins.shape[-1]

    def sample(self, num_rays: int) -> "RayBundle":
        """cameras/rays.py:217-229: a random subset of the rays."""
        assert num_rays <= len(self)
        return self[random.sample(range(len(self)), k=num_rays)]

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        """cameras/rays.py:238-249."""
        return self.flatten()[start_idx:end_idx]

    def get_ray_samples(self, bin_starts: Tensor, bin_ends: Tensor, spacing_starts: Optional[Tensor] = None, spacing_ends: Optional[Tensor] = None,
                        spacing_to_euclidean_fn: Optional[Callable] = None) -> RaySamples:
        """cameras/rays.py:251-295: [*batch, S] frustums; the bundle's origins / directions / pixel_area enter as [*batch, 1, .] and are
        broadcast (views) along the samples."""
        deltas = bin_ends - bin_starts
        cam = self.camera_indices[..., None, :] if self.camera_indices is not None else None
        shaped = self[..., None]
        fr = Frustums(origins=shaped.origins, directions=shaped.directions, starts=bin_starts, ends=bin_ends, pixel_area=shaped.pixel_area)
        return RaySamples(frustums=fr, camera_indices=cam, deltas=deltas, spacing_starts=spacing_starts, spacing_ends=spacing_ends,
                          spacing_to_euclidean_fn=spacing_to_euclidean_fn, metadata=shaped.metadata,
                          times=None if self.times is None else self.times[..., None, :])


def ray_samples_from_level(bundle: RayBundle, s_bins: Tensor, e_bins: Tensor, nears: Tensor, fars: Tensor) -> RaySamples:
    """Wrap one level of the engine's dense bins as the reference's RaySamples (views only, no copies)."""

    def s2e(x: Tensor) -> Tensor:  # spacing_to_euclidean_fn closure (ray_samplers.py:113-118), element-wise in torch for API users
        sp = lambda v: torch.where(v < 1, v / 2, 1 - 1 / (2 * v))  # noqa: E731
        inv = lambda v: torch.where(v < 0.5, 2 * v, 1 / (2 - 2 * v))  # noqa: E731
        n, f = nears.reshape(-1, 1), fars.reshape(-1, 1)
        return inv(x * sp(f) + (1 - x) * sp(n))

    rs = bundle.get_ray_samples(bin_starts=e_bins[..., :-1, None], bin_ends=e_bins[..., 1:, None], spacing_starts=s_bins[..., :-1, None],
                                spacing_ends=s_bins[..., 1:, None], spacing_to_euclidean_fn=s2e)
    rs.s_bins, rs.e_bins = s_bins, e_bins
    return rs


class LazyRaySamples:
    """One level of the engine's dense bins that BECOMES the reference's RaySamples on first use of any of its fields.  The training loop
    only ever reads `s_bins` / `e_bins` from the entries of `ray_samples_list` (the loss kernels take the dense bins); building the full
    Frustums / RaySamples views for three levels costs ~60 tensor operations per iteration on the host, so they are built on demand."""

    def __init__(self, bundle, s_bins: Tensor, e_bins: Tensor, nears: Tensor, fars: Tensor):
        """bundle: the RayBundle of the level's rays, or a callable that builds it (the bundle's broadcast-on-construct is host time, too)."""
        self.s_bins, self.e_bins = s_bins, e_bins
        self._args = (bundle, nears, fars)
        self._real: Optional[RaySamples] = None

    def materialize(self) -> RaySamples:
        if self._real is None:
            bundle, nears, fars = self._args
            if not isinstance(bundle, RayBundle):
                bundle = bundle()
            self._real = ray_samples_from_level(bundle, self.s_bins, self.e_bins, nears, fars)
        return self._real

    def __getattr__(self, name):  # only reached for attributes not set in __init__: the RaySamples fields and methods
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)
