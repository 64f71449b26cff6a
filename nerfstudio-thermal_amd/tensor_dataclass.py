"""Batched dataclasses of tensors: the semantics of the reference's TensorDataclass (utils/tensor_dataclass.py:27-350) that RayBundle,
RaySamples and Frustums (cameras/rays.py:32-295) are built on.

A subclass is a @dataclass whose tensor fields share a BATCH shape: every tensor is `[*batch, d]` (one trailing feature dimension, or the
number given for its name in `_field_custom_dimensions`).  On construction the batch shapes of all tensor fields -- including tensors inside
dict fields and nested TensorDataclass fields -- are broadcast against each other and every field becomes a broadcast VIEW of the common
shape (no copy).  Indexing, reshape, flatten, broadcast_to and `.to(device)` then act on the batch dimensions of all fields at once and
return a new instance; non-tensor fields are carried along (deep-copied).  Index assignment is refused, as in the reference.
"""
from __future__ import annotations

import dataclasses
from copy import deepcopy
from typing import Any, Callable, Dict, Tuple

import numpy as np
import torch
from torch import Tensor


class TensorDataclass:
    _shape: Tuple[int, ...]
    _field_custom_dimensions: Dict[str, int] = {}  # field (or dict key) name -> trailing dimensions that are NOT batch dimensions (> 1)

    # ------------------------------------------------------------------------------------------------ construction
    def __post_init__(self) -> None:
        if not dataclasses.is_dataclass(self):
            raise TypeError("TensorDataclass must be a dataclass")
        for k, v in self._field_custom_dimensions.items():
            assert isinstance(v, int) and v > 1, f"Custom dimensions must be an integer greater than 1, since 1 is the default, received {k}: {v}"
        shapes = []
        self._collect_batch_shapes(self._fields_dict(), shapes)
        if not shapes:
            raise ValueError("TensorDataclass must have at least one tensor")
        first = shapes[0]
        if all(s == first for s in shapes):  # the usual case (every per-iteration bundle): nothing to broadcast, no views to build
            object.__setattr__(self, "_shape", first)
            return
        batch = tuple(torch.broadcast_shapes(*shapes))
        for name, value in self._expand_dict(self._fields_dict(), batch).items():
            object.__setattr__(self, name, value)
        object.__setattr__(self, "_shape", batch)

    def _fields_dict(self) -> Dict[str, Any]:
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}

    def _trailing(self, name: str) -> int:
        return self._field_custom_dimensions.get(name, 1) if isinstance(self._field_custom_dimensions, dict) else 1

    def _collect_batch_shapes(self, items: Dict[str, Any], out: list) -> None:
        for name, v in items.items():
            if isinstance(v, Tensor):
                out.append(tuple(v.shape[: v.dim() - self._trailing(name)]))
            elif isinstance(v, TensorDataclass):
                out.append(tuple(v.shape))
            elif isinstance(v, dict):
                self._collect_batch_shapes(v, out)

    def _expand_dict(self, items: Dict[str, Any], batch: Tuple[int, ...]) -> Dict[str, Any]:
        out = {}
        for name, v in items.items():
            if isinstance(v, Tensor):
                t = self._trailing(name)
                out[name] = v.broadcast_to((*batch, *v.shape[v.dim() - t:]))
            elif isinstance(v, TensorDataclass):
                out[name] = v.broadcast_to(batch)
            elif isinstance(v, dict):
                out[name] = self._expand_dict(v, batch)
            else:
                out[name] = v
        return out

    # ------------------------------------------------------------------------------------------------ the one traversal everything else uses
    def _rebuild(self, on_tensor: Callable[[Tensor, int], Tensor], on_dataclass: Callable[["TensorDataclass"], "TensorDataclass"]):
        """New instance with `on_tensor(value, trailing_dims)` applied to every tensor leaf and `on_dataclass` to nested TensorDataclasses."""

        def walk(items: Dict[str, Any], top: bool) -> Dict[str, Any]:
            out = {}
            for name, v in items.items():
                if v is None:
                    continue
                if isinstance(v, TensorDataclass):
                    out[name] = on_dataclass(v)
                elif isinstance(v, Tensor):
                    # (the reference honours custom dimensions for direct fields only; inside dict fields a tensor has one trailing dimension)
                    out[name] = on_tensor(v, self._trailing(name) if top else 1)
                elif isinstance(v, dict):
                    out[name] = walk(v, False)
                else:
                    out[name] = deepcopy(v)
            return out

        return dataclasses.replace(self, **walk(self._fields_dict(), True))

    # ------------------------------------------------------------------------------------------------ batch-shape API
    @property
    def shape(self) -> Tuple[int, ...]:
        return self._shape

    @property
    def size(self) -> int:
        return 1 if len(self._shape) == 0 else int(np.prod(self._shape))

    @property
    def ndim(self) -> int:
        return len(self._shape)

    def __len__(self) -> int:
        if len(self._shape) == 0:
            raise TypeError("len() of a 0-d tensor")
        return self._shape[0]

    def __bool__(self) -> bool:
        if len(self) == 0:
            raise ValueError(f"The truth value of {self.__class__.__name__} when `len(x) == 0` is ambiguous. Use `len(x)` or `x is not None`.")
        return True

    def __getitem__(self, indices):
        if isinstance(indices, Tensor):
            return self._rebuild(lambda x, t: x[indices], lambda d: d[indices])
        if isinstance(indices, (int, slice, type(Ellipsis))):
            indices = (indices,)
        if isinstance(indices, list):  # RayBundle.sample indexes with a list of ray numbers
            return self._rebuild(lambda x, t: x[indices], lambda d: d[indices])
        assert isinstance(indices, tuple)
        return self._rebuild(lambda x, t: x[indices + (slice(None),) * t], lambda d: d[indices])

    def __setitem__(self, indices, value):
        raise RuntimeError("Index assignment is not supported for TensorDataclass")

    def reshape(self, shape):
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(shape)
        return self._rebuild(lambda x, t: x.reshape((*shape, *x.shape[x.dim() - t:])), lambda d: d.reshape(shape))

    def flatten(self):
        return self.reshape((-1,))

    def broadcast_to(self, shape):
        shape = tuple(shape)
        return self._rebuild(lambda x, t: x.broadcast_to((*shape, *x.shape[x.dim() - t:])), lambda d: d.broadcast_to(shape))

    def to(self, device):
        return self._rebuild(lambda x, t: x.to(device), lambda d: d.to(device))

    def pin_memory(self):
        return self._rebuild(lambda x, t: x.pin_memory(), lambda d: d.pin_memory())
