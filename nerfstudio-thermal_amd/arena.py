"""Flat fp32 parameter arena.

All trainable tensors live in ONE contiguous device buffer, grouped by optimiser group, with three sibling
buffers (gradients, Adam exp_avg, Adam exp_avg_sq).  The nn.Parameters the model exposes (under the reference's
state_dict names) are views into it, so
  * the optimiser is one fused kernel per group over a contiguous range (engine/optimizers.py:73-210 builds one
    torch.optim.Adam per group),
  * the data-parallel gradient exchange is ONE RCCL all-reduce over the live range instead of DDP's 25 MB buckets
    (pipelines/base_pipeline.py:282), and
  * zeroing the gradients is one memset.
Layout order: [proposal_networks | fields | camera_opt | proposal_networks_thermal | fields_thermal | camera_opt_thermal].
In shared mode the thermal proposal nets / thermal pose exist (the reference constructs them unconditionally,
models/thermal_nerfacto.py:138-186) but belong to no optimiser group: they sit after the live range.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

from .config import ThermalNerfactoModelConfig


def field_shapes(prefix: str, cfg: ThermalNerfactoModelConfig, num_images: int, channels: int) -> "OrderedDict[str, Tuple[int, ...]]":
    T = 2**cfg.log2_hashmap_size
    F = cfg.features_per_level
    din = 16 + 15 + cfg.appearance_embed_dim
    return OrderedDict(
        [
            # the table first: the small tensors (MLPs, appearance embedding) then form ONE contiguous tail of the group, next to the pose
            # parameters of the following group -- a single small collective in data-parallel runs
            (f"{prefix}.mlp_base.model.0.hash_table", (T * cfg.num_levels, F)),
            (f"{prefix}.embedding_appearance.embedding.weight", (num_images, cfg.appearance_embed_dim)),
            (f"{prefix}.mlp_base.model.1.layers.0.weight", (cfg.hidden_dim, cfg.num_levels * F)),
            (f"{prefix}.mlp_base.model.1.layers.0.bias", (cfg.hidden_dim,)),
            (f"{prefix}.mlp_base.model.1.layers.1.weight", (16, cfg.hidden_dim)),
            (f"{prefix}.mlp_base.model.1.layers.1.bias", (16,)),
            (f"{prefix}.mlp_head.layers.0.weight", (cfg.hidden_dim_color, din)),
            (f"{prefix}.mlp_head.layers.0.bias", (cfg.hidden_dim_color,)),
            (f"{prefix}.mlp_head.layers.1.weight", (cfg.hidden_dim_color, cfg.hidden_dim_color)),
            (f"{prefix}.mlp_head.layers.1.bias", (cfg.hidden_dim_color,)),
            (f"{prefix}.mlp_head.layers.2.weight", (channels, cfg.hidden_dim_color)),
            (f"{prefix}.mlp_head.layers.2.bias", (channels,)),
        ]
    )


def prop_shapes(prefix: str, cfg: ThermalNerfactoModelConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for i in range(cfg.num_proposal_iterations):
        a = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
        T = 2 ** a["log2_hashmap_size"]
        L, H = a["num_levels"], a["hidden_dim"]
        out[f"{prefix}.{i}.mlp_base.0.hash_table"] = (T * L, cfg.features_per_level)
        out[f"{prefix}.{i}.mlp_base.1.layers.0.weight"] = (H, L * cfg.features_per_level)
        out[f"{prefix}.{i}.mlp_base.1.layers.0.bias"] = (H,)
        out[f"{prefix}.{i}.mlp_base.1.layers.1.weight"] = (1, H)
        out[f"{prefix}.{i}.mlp_base.1.layers.1.bias"] = (1,)
    return out


class ParamArena:
    ALIGN = 64  # floats (256 B): every tensor starts on a 256-byte boundary
    _arenas: list = []  # weak references to the live arenas, so that an optimiser can find the arena a Parameter is a view of

    @classmethod
    def owner_of(cls, p: torch.Tensor):
        ptr = p.data_ptr()
        for ref in cls._arenas:
            a = ref()
            if a is None:
                continue
            lo = a.params.data_ptr()
            if p.device == a.params.device and lo <= ptr < lo + a.params.numel() * 4:
                return a
        return None

    def __init__(self, cfg: ThermalNerfactoModelConfig, num_images: int, device):
        separate = cfg.density_mode == "separate"
        groups: "OrderedDict[str, OrderedDict]" = OrderedDict()
        groups["proposal_networks"] = prop_shapes("proposal_networks", cfg)
        groups["fields"] = field_shapes("field", cfg, num_images, 3 + (0 if separate else 1))
        groups["camera_opt"] = OrderedDict([("camera_optimizer.pose_adjustment", (num_images, 6))]) if cfg.camera_optimizer.mode != "off" else OrderedDict()
        groups["proposal_networks_thermal"] = prop_shapes("proposal_networks_thermal", cfg)
        groups["fields_thermal"] = field_shapes("field_thermal", cfg, num_images, 1) if separate else OrderedDict()
        groups["camera_opt_thermal"] = (
            OrderedDict([("camera_optimizer_thermal.pose_adjustment", (num_images, 6))]) if cfg.camera_optimizer_thermal.mode != "off" else OrderedDict()
        )
        self.optimised_groups: List[str] = ["proposal_networks", "fields", "camera_opt"]
        if separate:
            self.optimised_groups += ["proposal_networks_thermal", "fields_thermal", "camera_opt_thermal"]
        self.optimised_groups = [g for g in self.optimised_groups if groups[g]]
        self.layout: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        self.group_range: Dict[str, Tuple[int, int]] = {}
        self.group_keys: Dict[str, List[str]] = {}
        off = 0
        for gname, shapes in groups.items():
            start = off
            for name, shape in shapes.items():
                n = 1
                for s in shape:
                    n *= s
                self.layout[name] = (off, tuple(shape))
                off += ((n + self.ALIGN - 1) // self.ALIGN) * self.ALIGN
            self.group_range[gname] = (start, off)
            self.group_keys[gname] = list(shapes.keys())
        self.total = off
        live = [self.group_range[g] for g in self.optimised_groups]
        self.live_range = (min(a for a, _ in live), max(b for _, b in live))
        self.device = torch.device(device)
        self.params = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        # True while the gradient buffer is known to be all zero (freshly allocated, zero_grad(), or consumed by an optimiser launch that zeroes
        # behind its read).  EVERY writer outside those clears it -- the fused backward, the drop-in path's autograd nodes, gradients copied in
        # by HipFusedAdam -- so that the fused step never starts on another path's leftovers.
        self.grads_clean = True
        self.exp_avg = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        ParamArena._arenas[:] = [r for r in ParamArena._arenas if r() is not None]
        ParamArena._arenas.append(weakref.ref(self))

    # ---------------------------------------------------------------- views
    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.layout[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off : off + n].view(shape)

    def view(self, name: str) -> torch.Tensor:
        return self._view(self.params, name)

    def grad_view(self, name: str) -> torch.Tensor:
        """a FRESH view of the parameter's slice of the gradient buffer (one as_strided: the drop-in backward hands out ~30 of them per step)"""
        spec = self.__dict__.get("_grad_view_spec")
        if spec is None:
            spec = self._grad_view_spec = {}
            for n, (off, shape) in self.layout.items():
                stride, k = [], 1
                for d in reversed(shape):
                    stride.append(k)
                    k *= d
                spec[n] = (tuple(shape), tuple(reversed(stride)), off)
        shape, stride, off = spec[name]
        return self.grads.as_strided(shape, stride, self.grads.storage_offset() + off)

    def grad_ptrs(self) -> Dict[str, int]:
        self.grad_ptr(next(iter(self.layout)))
        return self._grad_ptrs

    def grad_ptr(self, name: str) -> int:
        """device address of a parameter's slice of the gradient buffer (to recognise a `.grad` that aliases it without building a view)"""
        base = self.__dict__.get("_grad_ptr_base")
        if base != self.grads.data_ptr():
            self._grad_ptr_base = base = self.grads.data_ptr()
            self._grad_ptrs = {n: base + 4 * off for n, (off, _) in self.layout.items()}
        return self._grad_ptrs[name]

    def names(self) -> List[str]:
        return list(self.layout.keys())

    def load(self, tensors: Dict[str, torch.Tensor]) -> None:
        """Copy host/device tensors (reference state_dict names) into the arena."""
        with torch.no_grad():
            for name in self.layout:
                if name in tensors:
                    self.view(name).copy_(torch.as_tensor(tensors[name]).to(self.device, torch.float32).reshape(self.layout[name][1]))

    def zero_grad(self) -> None:
        self.grads.zero_()
        self.grads_clean = True

    def num_optimised(self) -> int:
        n = 0
        for g in self.optimised_groups:
            for k in self.group_keys[g]:
                m = 1
                for s in self.layout[k][1]:
                    m *= s
                n += m
        return n
