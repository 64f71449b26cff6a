"""Boundary types of the hot path: RayBundle / RaySamples / Frustums with the reference's field names and methods
(cameras/rays.py:32-295), minus the generic TensorDataclass machinery: the HIP path keeps rays flat [N, ...] and a level's samples
dense [N, S, ...].  `RaySamples.get_weights` and `Frustums.get_positions` run on the device through the C ABI.
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from . import ops


def _map(obj, fn):
    kw = {}
    for f in fields(obj):
        v = getattr(obj, f.name)
        if isinstance(v, Tensor):
            kw[f.name] = fn(v)
        elif isinstance(v, dict):
            kw[f.name] = {k: (fn(x) if isinstance(x, Tensor) else x) for k, x in v.items()}
        elif hasattr(v, "__dataclass_fields__"):
            kw[f.name] = _map(v, fn)
        else:
            kw[f.name] = v
    return type(obj)(**kw)


@dataclass
class Frustums:
    origins: Tensor  # [N,1,3] (broadcast along samples) or [N,S,3]
    directions: Tensor
    starts: Tensor  # [N,S,1]
    ends: Tensor
    pixel_area: Tensor
    offsets: Optional[Tensor] = None

    @property
    def shape(self):
        return self.starts.shape[:-1]

    def get_positions(self) -> Tensor:
        """cameras/rays.py:49-58 (element-wise; evaluated by torch on the device: it is not on the fused path, which recomputes positions in-kernel)."""
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        if self.offsets is not None:
            pos = pos + self.offsets
        return pos

    def get_start_positions(self) -> Tensor:
        return self.origins + self.directions * self.starts


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[Tensor] = None
    deltas: Optional[Tensor] = None
    spacing_starts: Optional[Tensor] = None
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None
    # dense level tensors kept for the kernels (not reference fields)
    s_bins: Optional[Tensor] = field(default=None, repr=False)
    e_bins: Optional[Tensor] = field(default=None, repr=False)

    @property
    def shape(self):
        return self.frustums.shape

    def get_weights(self, densities: Tensor) -> Tensor:
        """RaySamples.get_weights (cameras/rays.py:128-150) -> tn_weights_fwd.  densities [N,S,1] -> weights [N,S,1]."""
        if self.e_bins is None:
            self.e_bins = torch.cat([self.frustums.starts[..., 0], self.frustums.ends[..., -1:, 0]], dim=-1).contiguous()
        w, _ = ops.weights_fwd(self.e_bins, densities[..., 0].contiguous())
        return w.unsqueeze(-1)


@dataclass
class RayBundle:
    origins: Tensor
    directions: Tensor
    pixel_area: Tensor
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def __len__(self) -> int:
        return self.origins.numel() // self.origins.shape[-1]

    def set_camera_indices(self, camera_index: int) -> None:
        self.camera_indices = torch.ones_like(self.origins[..., 0:1]).long() * camera_index

    def to(self, device) -> "RayBundle":
        return _map(self, lambda t: t.to(device))

    def flatten(self) -> "RayBundle":
        n = len(self)
        return _map(self, lambda t: t.reshape(n, t.shape[-1]) if t.dim() >= 2 and t.numel() // max(t.shape[-1], 1) == n else t)

    def __getitem__(self, idx) -> "RayBundle":
        return _map(self, lambda t: t[idx])

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        """cameras/rays.py:238-249."""
        return self.flatten()[start_idx:end_idx]

    def get_ray_samples(self, bin_starts: Tensor, bin_ends: Tensor, spacing_starts: Optional[Tensor] = None, spacing_ends: Optional[Tensor] = None,
                        spacing_to_euclidean_fn: Optional[Callable] = None) -> RaySamples:
        """cameras/rays.py:251-295: [N,S] frustums with the bundle's origins/directions/pixel_area broadcast as [N,1,.] views."""
        fr = Frustums(origins=self.origins[..., None, :], directions=self.directions[..., None, :], starts=bin_starts, ends=bin_ends,
                      pixel_area=self.pixel_area[..., None, :])
        cam = self.camera_indices[..., None] if self.camera_indices is not None else None
        return RaySamples(frustums=fr, camera_indices=cam, deltas=bin_ends - bin_starts, spacing_starts=spacing_starts, spacing_ends=spacing_ends,
                          spacing_to_euclidean_fn=spacing_to_euclidean_fn, metadata={k: v[..., None, :] for k, v in self.metadata.items()},
                          times=None if self.times is None else self.times[..., None])


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

    def __init__(self, bundle: RayBundle, s_bins: Tensor, e_bins: Tensor, nears: Tensor, fars: Tensor):
        self.s_bins, self.e_bins = s_bins, e_bins
        self._args = (bundle, nears, fars)
        self._real: Optional[RaySamples] = None

    def materialize(self) -> RaySamples:
        if self._real is None:
            bundle, nears, fars = self._args
            self._real = ray_samples_from_level(bundle, self.s_bins, self.e_bins, nears, fars)
        return self._real

    def __getattr__(self, name):  # only reached for attributes not set in __init__: the RaySamples fields and methods
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)
