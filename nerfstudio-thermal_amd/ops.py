"""Tensor-level wrappers over the C ABI: torch CUDA tensors in, torch CUDA tensors out.

PyTorch is plumbing here (device memory + the current HIP stream); all arithmetic runs in
libthermal_nerf_hip.so.  Every function validates device / dtype / contiguity / shape on the host before a
kernel is launched (a wrong shape must never reach a hand-written kernel).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from ._lib import TnField, TnGrid, TnPropNet, check

TN_MAX_SAMPLES = _lib.TN_MAX_SAMPLES


def level_resolutions(num_levels: int, min_res: int, max_res: int) -> List[float]:
    """floor(min_res * growth**level) exactly as the reference evaluates it (field_components/encodings.py:343-345):
    numpy-float64 growth, torch int64 arange -> float32 pow; the default main grid tops out at 2047."""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * growth**levels).to(torch.float32).tolist()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_STRIDES: dict = {}


def _contig_strides(shape: Tuple[int, ...]) -> Tuple[int, ...]:
    st = _STRIDES.get(shape)
    if st is None:
        out, k = [], 1
        for d in reversed(shape):
            out.append(k)
            k *= d
        st = _STRIDES[shape] = tuple(reversed(out))
    return st


def _stream() -> C.c_void_p:
    """torch's current stream as a raw hipStream_t (the private accessor costs ~0.3 us, torch.cuda.current_stream() ~10 us: a step makes
    ~50 launches)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: Optional[Tensor], name: str, shape: Optional[Tuple[int, ...]] = None, optional: bool = False):
    if t is None:
        if optional:
            return None
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA(HIP) tensor: the thermal-nerfacto path has no CPU fallback")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return C.c_void_p(t.data_ptr())


def _i64(t: Tensor, name: str, shape: Tuple[int, ...]):
    if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must be a contiguous CUDA int64 tensor of shape {shape}, got {t.dtype} {tuple(t.shape)}")
    return C.c_void_p(t.data_ptr())


def _grid_struct(table: Tensor, grad: Optional[Tensor], num_levels: int, log2_hashmap_size: int, res: Sequence[float],
                 nonfinite_flag: Optional[Tensor] = None, grad_is_zero: bool = False) -> TnGrid:
    """nonfinite_flag: 1-element float device tensor the table-gradient scatter raises on an inf / NaN entry (TnGrid.nonfinite_flag), or None.
    grad_is_zero: the caller's promise that `grad` holds zeros when a scatter starts (TnGrid.table_grad_is_zero: the fold stores instead of adding)."""
    T = 2**log2_hashmap_size
    g = TnGrid()
    g.table_grad_is_zero = 1 if (grad_is_zero and grad is not None) else 0
    g.nonfinite_flag = _f32(nonfinite_flag, "nonfinite_flag", (1,), optional=True) if (nonfinite_flag is not None and grad is not None) else None
    g.table = _f32(table, "hash_table", (num_levels * T, 2))
    g.table_grad = _f32(grad, "hash_table.grad", (num_levels * T, 2), optional=True)
    g.num_levels = num_levels
    g.log2_hashmap_size = log2_hashmap_size
    if len(res) != num_levels or num_levels > _lib.TN_MAX_LEVELS:
        raise ValueError("bad level resolutions")
    for i in range(_lib.TN_MAX_LEVELS):
        g.res[i] = float(res[i]) if i < num_levels else 0.0
    return g


@dataclass
class PropNetParams:
    """HashMLPDensityField parameters (fields/density_fields.py:34-118); tensors are views of the caller's storage."""

    table: Tensor
    w0: Tensor
    b0: Tensor
    w1: Tensor
    b1: Tensor
    num_levels: int
    log2_hashmap_size: int
    res: List[float]
    grads: Optional[dict] = None  # same keys -> gradient tensors

    def __setattr__(self, name, value):  # a new tensor in any field: the checked C struct is rebuilt at the next call
        self.__dict__.pop("_cs", None)
        object.__setattr__(self, name, value)

    def cstruct(self, need_grad: bool = False) -> TnPropNet:
        g = self.grads or {}
        if need_grad and not g:
            raise ValueError("gradient buffers required")
        # the parameters are views of the arena: their addresses do not change from step to step, so the checked struct is built once
        nf = self.__dict__.get("nonfinite_flag")  # set by the engine: the optimiser group's found_inf entry (DeviceGradScaler), or absent
        # (key: the table's addresses stand for all of them -- every tensor is a view of the same arena; assigning a field drops the cached
        # struct, see __setattr__ -- 25 data_ptr() calls per struct and step were ~10 us each on the path to the two library calls)
        gt = g.get("table")
        key = (self.table.data_ptr(), gt.data_ptr() if gt is not None else 0, len(g), nf.data_ptr() if nf is not None else 0,
               bool(self.__dict__.get("grad_is_zero", False)))
        cache = self.__dict__.get("_cs")
        if cache is None:
            cache = self.__dict__["_cs"] = {}
        s = cache.get(key)
        if s is None:
            if len(cache) >= 8:  # (addresses moved, flags toggled: a handful of variants at most)
                cache.clear()
            s = cache[key] = self._build_cstruct(g)
        return s

    def _build_cstruct(self, g) -> TnPropNet:
        s = TnPropNet()
        s.grid = _grid_struct(self.table, g.get("table"), self.num_levels, self.log2_hashmap_size, self.res, self.__dict__.get("nonfinite_flag"),
                              bool(self.__dict__.get("grad_is_zero", False)))
        H, F = 16, self.num_levels * 2
        s.w0, s.b0 = _f32(self.w0, "w0", (H, F)), _f32(self.b0, "b0", (H,))
        s.w1, s.b1 = _f32(self.w1, "w1", (1, H)), _f32(self.b1, "b1", (1,))
        s.gw0, s.gb0 = _f32(g.get("w0"), "gw0", (H, F), True), _f32(g.get("b0"), "gb0", (H,), True)
        s.gw1, s.gb1 = _f32(g.get("w1"), "gw1", (1, H), True), _f32(g.get("b1"), "gb1", (1,), True)
        return s


_FIELD_KEYS = ("w0", "b0", "w1", "b1", "hw0", "hb0", "hw1", "hb1", "hw2", "hb2", "emb")


@dataclass
class FieldParams:
    """ThermalNerfactoField parameters (fields/thermal_nerfacto_field.py:37-99)."""

    table: Tensor
    w0: Tensor
    b0: Tensor
    w1: Tensor
    b1: Tensor
    hw0: Tensor
    hb0: Tensor
    hw1: Tensor
    hb1: Tensor
    hw2: Tensor
    hb2: Tensor
    emb: Tensor
    num_levels: int
    log2_hashmap_size: int
    res: List[float]
    num_channels: int
    grads: Optional[dict] = None
    _ws: dict = field(default_factory=dict, repr=False)

    def shapes(self):
        C_, I = self.num_channels, self.emb.shape[0]
        return {"w0": (64, 32), "b0": (64,), "w1": (16, 64), "b1": (16,), "hw0": (64, 63), "hb0": (64,), "hw1": (64, 64),
                "hb1": (64,), "hw2": (C_, 64), "hb2": (C_,), "emb": (I, 32)}

    def __setattr__(self, name, value):  # a new tensor in any field: the checked C struct is rebuilt at the next call
        self.__dict__.pop("_cs", None)
        object.__setattr__(self, name, value)

    def cstruct(self, need_grad: bool = False) -> TnField:
        if self.num_levels * 2 != 32:
            raise ValueError("the fused main-field kernels are built for 16 levels x 2 features")
        g = self.grads or {}
        if need_grad and not g:
            raise ValueError("gradient buffers required")
        nf = self.__dict__.get("nonfinite_flag")
        gt = g.get("table")  # (as PropNetParams.cstruct: the table's addresses stand for the arena's; __setattr__ drops the cache)
        key = (self.table.data_ptr(), gt.data_ptr() if gt is not None else 0, len(g), nf.data_ptr() if nf is not None else 0,
               bool(self.__dict__.get("grad_is_zero", False)))
        cache = self.__dict__.get("_cs")
        if cache is None:
            cache = self.__dict__["_cs"] = {}
        s = cache.get(key)
        if s is None:
            if len(cache) >= 8:
                cache.clear()
            s = cache[key] = self._build_cstruct(g)
        return s

    def _build_cstruct(self, g) -> TnField:
        s = TnField()
        s.grid = _grid_struct(self.table, g.get("table"), self.num_levels, self.log2_hashmap_size, self.res, self.__dict__.get("nonfinite_flag"),
                              bool(self.__dict__.get("grad_is_zero", False)))
        for k, shp in self.shapes().items():
            setattr(s, k, _f32(getattr(self, k), k, shp))
            setattr(s, "g" + k, _f32(g.get(k), "g" + k, shp, optional=True))
        s.num_channels = self.num_channels
        s.num_images = self.emb.shape[0]
        return s

    def workspace(self, num_points: int, training: bool, tag: str = "main") -> Tensor:
        """Device scratch for tn_field_* (packed weights + saved activations), grown on demand and reused.
        `tag` names independent evaluations whose saved activations must coexist (e.g. the cross-evaluated densities)."""
        need = int(_lib.load().tn_field_workspace_bytes(num_points, 1 if training else 0))
        key = "ws_" + tag
        cur = self._ws.get(key)
        if cur is None or cur.numel() < need or cur.device != self.table.device:
            cur = torch.empty(need, dtype=torch.uint8, device=self.table.device)
            self._ws[key] = cur
        return cur


class UniformPool:
    """torch.rand in bulk: the per-step uniforms of the training loop (pixel sampler: [N/4,3]; sampler jitter: [3,N]) are a few KB each, and a
    launch for a few KB costs what a launch costs (~5 us on the step's serial chain).  take(shape) hands out fresh, never-reused slices of a
    buffer drawn `steps` requests at a time by ONE torch.rand call; every refill is a new allocation, so earlier slices stay valid."""

    def __init__(self, device, steps: int = 32):
        self.device, self.steps = device, int(steps)
        self._buf: Optional[Tensor] = None
        self._pos = 0

    def take(self, shape) -> Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        span = (n + 63) // 64 * 64  # 256-byte aligned slices
        if self._buf is None or self._pos + span > self._buf.numel():
            self._buf = torch.rand(max(span * self.steps, span), device=self.device)
            self._pos = 0
        out = self._buf[self._pos:self._pos + n].view(*shape)
        self._pos += span
        return out


# ------------------------------------------------------------------------------------------------ N2 pixel sampling
@dataclass
class ImageCache:
    """The cached training images resident in HBM (what CacheDataloader keeps on the host for the reference, data/utils/dataloaders.py:39-162):
    all images back to back in one fp32 buffer, in batch order."""

    buffer: Tensor        # [sum H_i*W_i*3] fp32
    offsets: Tensor       # [num_images] int64, float offset of image i
    heights: Tensor       # [num_images] int32
    widths: Tensor        # [num_images] int32
    is_thermal: Tensor    # [num_images] fp32, by batch position
    image_idx: Tensor     # [num_images] int64, dataset (camera) index of each batch position

    @staticmethod
    def build(images: Sequence[Tensor], is_thermal: Tensor, image_idx: Tensor, device) -> "ImageCache":
        offs, hs, ws, tot = [], [], [], 0
        for im in images:
            if im.dim() != 3 or im.shape[2] != 3:
                raise ValueError("images must be [H,W,3]")
            offs.append(tot)
            hs.append(im.shape[0])
            ws.append(im.shape[1])
            tot += im.numel()
        buf = torch.cat([im.reshape(-1).to(torch.float32) for im in images]).to(device)
        return ImageCache(buf, torch.tensor(offs, dtype=torch.int64, device=device), torch.tensor(hs, dtype=torch.int32, device=device),
                          torch.tensor(ws, dtype=torch.int32, device=device), is_thermal.to(device, torch.float32).contiguous(),
                          image_idx.to(device, torch.int64).contiguous())


def sample_pixels(cache: ImageCache, num_rays: int, u: Tensor, patch_size: int = 2, want_camera_indices: bool = False):
    """PatchPixelSampler.sample + ground-truth gather on the device -> ray_indices [N,3] int64 (camera,row,col), image [N,3], is_thermal [N]
    (+ camera_indices [N] int64 with want_camera_indices=True: ray_indices[:,0] as a contiguous vector).
    u [num_rays / patch^2, 3]: the uniforms the reference would draw with torch.rand, image after image."""
    n_img = cache.offsets.shape[0]
    dev = cache.buffer.device
    if u.device != dev or u.dtype != torch.float32 or not u.is_contiguous() or tuple(u.shape) != (num_rays // (patch_size * patch_size), 3):
        raise ValueError(f"u must be a contiguous fp32 [{num_rays // (patch_size * patch_size)}, 3] tensor on {dev}")
    idx = torch.empty((num_rays, 3), dtype=torch.int64, device=dev)
    img = torch.empty((num_rays, 3), dtype=torch.float32, device=dev)
    is_th = torch.empty((num_rays,), dtype=torch.float32, device=dev)
    cam = torch.empty((num_rays,), dtype=torch.int64, device=dev) if want_camera_indices else None
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    check(_lib.load().tn_sample_pixels(p(cache.buffer), p(cache.offsets), p(cache.heights), p(cache.widths), p(cache.is_thermal), p(cache.image_idx),
                                       n_img, p(u), num_rays, patch_size, p(idx), p(img), p(is_th), p(cam) if cam is not None else None, _stream()),
          "tn_sample_pixels")
    return (idx, img, is_th, cam) if want_camera_indices else (idx, img, is_th)


def sample_rays(cache: ImageCache, num_rays: int, u: Tensor, cameras: dict, patch_size: int = 2, want_pixel_area: bool = True, with_bundle_extras: bool = False):
    """sample_pixels + raygen in one launch (tn_sample_rays) -> origins [N,3], directions [N,3], camera_indices [N] int64, image [N,3],
    is_thermal [N], ray_indices [N,3] (+ pixel_area [N,1], directions_norm [N,1] with with_bundle_extras=True: the rest of the reference's
    RayBundle, cameras/cameras.py:904-928).  cameras: c2w [C,3,4], fx, fy, cx, cy [C], optional distortion [C,6].
    want_pixel_area=False: the bundle's pixel_area (which thermal-nerfacto never reads) is not computed -- two of the three undistortions per ray."""
    n_img = cache.offsets.shape[0]
    dev = cache.buffer.device
    if u.device != dev or u.dtype != torch.float32 or not u.is_contiguous() or tuple(u.shape) != (num_rays // (patch_size * patch_size), 3):
        raise ValueError(f"u must be a contiguous fp32 [{num_rays // (patch_size * patch_size)}, 3] tensor on {dev}")
    idx = torch.empty((num_rays, 3), dtype=torch.int64, device=dev)
    cam = torch.empty((num_rays,), dtype=torch.int64, device=dev)
    img, is_th, o, d, area, nrm = (torch.empty((num_rays, 3), device=dev), torch.empty((num_rays,), device=dev), torch.empty((num_rays, 3), device=dev),
                                   torch.empty((num_rays, 3), device=dev), torch.empty((num_rays, 1), device=dev), torch.empty((num_rays, 1), device=dev))
    c2w = cameras["c2w"]
    Cn = c2w.shape[0]
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    check(_lib.load().tn_sample_rays(p(cache.buffer), p(cache.offsets), p(cache.heights), p(cache.widths), p(cache.is_thermal), p(cache.image_idx),
                                     n_img, p(u), num_rays, patch_size, p(idx), p(img), p(is_th), p(cam),
                                     _f32(c2w, "c2w", (Cn, 3, 4)), _f32(cameras["fx"], "fx", (Cn,)), _f32(cameras["fy"], "fy", (Cn,)),
                                     _f32(cameras["cx"], "cx", (Cn,)), _f32(cameras["cy"], "cy", (Cn,)),
                                     _f32(cameras.get("distortion"), "distortion", (Cn, 6), optional=True), Cn, p(o), p(d), p(area) if want_pixel_area else None,
                                     p(nrm), _stream()),
          "tn_sample_rays")
    if with_bundle_extras:
        return o, d, cam, img, is_th, idx, area, nrm
    return o, d, cam, img, is_th, idx


# A batch whose sampling has been handed to the NEXT tn_train_step: that call runs it in co-work blocks of its optimiser launch (TnTrainStep.next_sample);
# if no such call comes first, flush_pending_sample() launches it as a kernel of its own.  One slot per process: the training loop has one data manager.
_PENDING_SAMPLE: Optional[tuple] = None


def sample_rays_deferred(cache: ImageCache, num_rays: int, u: Tensor, cameras: dict, patch_size: int = 2):
    """The arguments of sample_rays as a TnSampleRays block WITHOUT a launch -> origins, directions, camera_indices, image, is_thermal, ray_indices,
    pixel_area, directions_norm (the tensors sample_rays(with_bundle_extras=True) returns), filled once the next tn_train_step (TrainStepCall.run
    picks the pending block up) or flush_pending_sample() has run.  An older block nobody took is launched first."""
    global _PENDING_SAMPLE
    flush_pending_sample()
    n_img = cache.offsets.shape[0]
    dev = cache.buffer.device
    if u.device != dev or u.dtype != torch.float32 or not u.is_contiguous() or tuple(u.shape) != (num_rays // (patch_size * patch_size), 3):
        raise ValueError(f"u must be a contiguous fp32 [{num_rays // (patch_size * patch_size)}, 3] tensor on {dev}")
    idx = torch.empty((num_rays, 3), dtype=torch.int64, device=dev)
    cam = torch.empty((num_rays,), dtype=torch.int64, device=dev)
    img, is_th, o, d, area, nrm = (torch.empty((num_rays, 3), device=dev), torch.empty((num_rays,), device=dev), torch.empty((num_rays, 3), device=dev),
                                   torch.empty((num_rays, 3), device=dev), torch.empty((num_rays, 1), device=dev), torch.empty((num_rays, 1), device=dev))
    c2w = cameras["c2w"]
    Cn = c2w.shape[0]
    st = _lib.TnSampleRays()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st.images, st.image_offsets, st.heights, st.widths = p(cache.buffer), p(cache.offsets), p(cache.heights), p(cache.widths)
    st.is_thermal, st.image_idx, st.num_images = p(cache.is_thermal), p(cache.image_idx), n_img
    st.u, st.num_rays, st.patch_size = p(u), num_rays, patch_size
    st.ray_indices, st.image, st.is_thermal_out, st.camera_indices = p(idx), p(img), p(is_th), p(cam)
    st.c2w = _f32(c2w, "c2w", (Cn, 3, 4))
    st.fx, st.fy = _f32(cameras["fx"], "fx", (Cn,)), _f32(cameras["fy"], "fy", (Cn,))
    st.cx, st.cy = _f32(cameras["cx"], "cx", (Cn,)), _f32(cameras["cy"], "cy", (Cn,))
    st.distortion = _f32(cameras.get("distortion"), "distortion", (Cn, 6), optional=True)
    st.num_cameras = Cn
    st.origins, st.directions, st.pixel_area, st.directions_norm = p(o), p(d), p(area), p(nrm)
    outs = (o, d, cam, img, is_th, idx, area, nrm)
    _PENDING_SAMPLE = (st, (u, cameras, cache) + outs)
    return outs


def flush_pending_sample() -> None:
    """launches a batch handed over by sample_rays_deferred that no tn_train_step has taken (tn_sample_rays_args, current stream)"""
    global _PENDING_SAMPLE
    pend = _PENDING_SAMPLE
    if pend is None:
        return
    _PENDING_SAMPLE = None
    check(_lib.load().tn_sample_rays_args(C.byref(pend[0]), _stream()), "tn_sample_rays_args")


# ------------------------------------------------------------------------------------------------ a1 / a4
def raygen(ray_indices: Tensor, c2w: Tensor, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor, distortion: Optional[Tensor]):
    N, Cn = ray_indices.shape[0], c2w.shape[0]
    dev = ray_indices.device
    o = torch.empty((N, 3), device=dev)
    d = torch.empty((N, 3), device=dev)
    area = torch.empty((N, 1), device=dev)
    nrm = torch.empty((N, 1), device=dev)
    check(_lib.load().tn_raygen(_i64(ray_indices, "ray_indices", (N, 3)), _f32(c2w, "c2w", (Cn, 3, 4)), _f32(fx, "fx", (Cn,)),
                                _f32(fy, "fy", (Cn,)), _f32(cx, "cx", (Cn,)), _f32(cy, "cy", (Cn,)),
                                _f32(distortion, "distortion", (Cn, 6), optional=True), Cn, N, _f32(o, "o"), _f32(d, "d"), _f32(area, "a"),
                                _f32(nrm, "n"), _stream()), "tn_raygen")
    return o, d, area, nrm


def _u8(t: Optional[Tensor], n: int):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.uint8 or tuple(t.shape) != (n,):
        raise ValueError("frozen mask must be CUDA uint8 [num_cameras]")
    return C.c_void_p(t.data_ptr())


def pose_apply_fwd(pose: Tensor, frozen: Optional[Tensor], cam: Tensor, origins: Tensor, directions: Tensor):
    N, Cn = origins.shape[0], pose.shape[0]
    o, d = torch.empty_like(origins), torch.empty_like(directions)
    check(_lib.load().tn_pose_apply_fwd(_f32(pose, "pose", (Cn, 6)), _u8(frozen, Cn), _i64(cam, "camera_indices", (N,)), _f32(origins, "origins", (N, 3)),
                                        _f32(directions, "directions", (N, 3)), N, Cn, _f32(o, "o"), _f32(d, "d"), _stream()), "tn_pose_apply_fwd")
    return o, d


def pose_spaced_bins(pose: Tensor, frozen: Optional[Tensor], cam: Tensor, origins: Tensor, directions: Tensor, nears: Tensor, fars: Tensor, S: int,
                     jitter: Optional[Tensor] = None):
    """pose_apply_fwd + spaced_bins in one launch (tn_pose_spaced_bins) -> origins', directions', s_bins [N,S+1], e_bins [N,S+1]."""
    N, Cn = origins.shape[0], pose.shape[0]
    o, d = torch.empty_like(origins), torch.empty_like(directions)
    s = torch.empty((N, S + 1), device=origins.device)
    e = torch.empty((N, S + 1), device=origins.device)
    check(_lib.load().tn_pose_spaced_bins(_f32(pose, "pose", (Cn, 6)), _u8(frozen, Cn), _i64(cam, "camera_indices", (N,)), _f32(origins, "origins", (N, 3)),
                                          _f32(directions, "directions", (N, 3)), N, Cn, _f32(o, "o"), _f32(d, "d"),
                                          _f32(_lin_table("spaced", S, origins.device), "lin"), _ray_scalar(jitter, "jitter", N, True),
                                          _ray_scalar(nears, "nears", N), _ray_scalar(fars, "fars", N), S, _f32(s, "s"), _f32(e, "e"), _stream()),
          "tn_pose_spaced_bins")
    return o, d, s, e


def pose_bwd_finish(pose: Tensor, frozen: Optional[Tensor], cam: Tensor, directions_in: Tensor, d_o: Tensor, d_d: Tensor, grad_pose: Tensor,
                    trans_pen: float, rot_pen: float, scale: float, reg_out: Tensor, loss_lines: Optional[Tensor] = None,
                    losses16: Optional[Tensor] = None, check_grads: Optional[Tensor] = None, check_ranges=None, found_inf: Optional[Tensor] = None,
                    pose_flag: int = 0) -> None:
    """pose_apply_bwd + camera_reg (+ losses_finish when loss_lines / losses16 are given) in one launch (tn_pose_bwd_finish).
    found_inf (+ check_grads, check_ranges = [(lo, hi, flag), ...] small ranges of the gradient arena, pose_flag): the same launch also raises
    GradScaler's per-group found_inf for the pose gradient and those ranges (tn_pose_bwd_finish_check)."""
    N, Cn = directions_in.shape[0], pose.shape[0]
    if found_inf is not None:
        rs = list(check_ranges or [])
        n = len(rs)
        offs = (C.c_int64 * max(n, 1))(*[r[0] for r in rs])
        cnts = (C.c_int64 * max(n, 1))(*[r[1] - r[0] for r in rs])
        fl = (C.c_int32 * max(n, 1))(*[int(r[2]) for r in rs])
        check(_lib.load().tn_pose_bwd_finish_check(_f32(pose, "pose", (Cn, 6)), _u8(frozen, Cn), _i64(cam, "camera_indices", (N,)),
                                                   _f32(directions_in, "directions", (N, 3)), _f32(d_o, "d_origins", (N, 3)),
                                                   _f32(d_d, "d_directions", (N, 3)), N, Cn, _f32(grad_pose, "grad_pose", (Cn, 6)),
                                                   _f32(loss_lines, "loss_lines", (LOSS_LINES, 16), True), _f32(losses16, "losses16", None, True),
                                                   float(trans_pen), float(rot_pen), float(scale), _f32(reg_out, "reg_out"),
                                                   _f32(check_grads, "grads", None, True), n, offs, cnts, fl, int(found_inf.numel()),
                                                   _f32(found_inf, "found_inf"), int(pose_flag), _stream()), "tn_pose_bwd_finish_check")
        return
    check(_lib.load().tn_pose_bwd_finish(_f32(pose, "pose", (Cn, 6)), _u8(frozen, Cn), _i64(cam, "camera_indices", (N,)),
                                         _f32(directions_in, "directions", (N, 3)), _f32(d_o, "d_origins", (N, 3)), _f32(d_d, "d_directions", (N, 3)),
                                         N, Cn, _f32(grad_pose, "grad_pose", (Cn, 6)),
                                         _f32(loss_lines, "loss_lines", (_lib.TN_LOSS_LINES, 16), optional=True), _f32(losses16, "losses16", optional=True),
                                         float(trans_pen), float(rot_pen), float(scale), _f32(reg_out, "reg_out"), _stream()), "tn_pose_bwd_finish")


def pose_apply_bwd(pose: Tensor, frozen: Optional[Tensor], cam: Tensor, directions_in: Tensor, d_o: Tensor, d_d: Tensor, grad_pose: Tensor):
    N, Cn = directions_in.shape[0], pose.shape[0]
    check(_lib.load().tn_pose_apply_bwd(_f32(pose, "pose", (Cn, 6)), _u8(frozen, Cn), _i64(cam, "camera_indices", (N,)),
                                        _f32(directions_in, "directions", (N, 3)), _f32(d_o, "d_origins", (N, 3)), _f32(d_d, "d_directions", (N, 3)),
                                        N, Cn, _f32(grad_pose, "grad_pose", (Cn, 6)), _stream()), "tn_pose_apply_bwd")


# ------------------------------------------------------------------------------------------------ samplers
_TABLES: dict = {}


def _lin_table(kind: str, S: int, device) -> Tensor:
    """torch.linspace tables the reference builds on the host (ray_samplers.py:100,319): computed by torch itself, uploaded once."""
    key = (kind, S, str(device))
    t = _TABLES.get(key)
    if t is None:
        if kind == "spaced":
            t = torch.linspace(0.0, 1.0, S + 1)
        else:
            nb = S + 1
            t = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
        t = t.to(device)
        _TABLES[key] = t
    return t


def _ray_scalar(t: Optional[Tensor], name: str, N: int, optional=False):
    if t is None and optional:
        return None
    if t.dim() == 2:
        t = t.reshape(-1)
    return _f32(t, name, (N,))


def spaced_bins(nears: Tensor, fars: Tensor, S: int, jitter: Optional[Tensor] = None):
    N = nears.shape[0]
    s = torch.empty((N, S + 1), device=nears.device)
    e = torch.empty((N, S + 1), device=nears.device)
    check(_lib.load().tn_spaced_bins(_f32(_lin_table("spaced", S, nears.device), "lin"), _ray_scalar(jitter, "jitter", N, True), _ray_scalar(nears, "nears", N),
                                     _ray_scalar(fars, "fars", N), N, S, _f32(s, "s"), _f32(e, "e"), _stream()), "tn_spaced_bins")
    return s, e


def pdf_resample(s_bins_prev: Tensor, weights_prev: Tensor, S: int, anneal: float, nears: Tensor, fars: Tensor, jitter: Optional[Tensor] = None):
    N, Sp = weights_prev.shape[0], weights_prev.shape[1]
    s = torch.empty((N, S + 1), device=nears.device)
    e = torch.empty((N, S + 1), device=nears.device)
    check(_lib.load().tn_pdf_resample(_f32(s_bins_prev, "s_bins_prev", (N, Sp + 1)), _f32(weights_prev, "weights_prev", (N, Sp)), Sp, float(anneal),
                                      _f32(_lin_table("pdf", S, nears.device), "u"), _ray_scalar(jitter, "jitter", N, True), _ray_scalar(nears, "nears", N),
                                      _ray_scalar(fars, "fars", N), N, S, _f32(s, "s"), _f32(e, "e"), _stream()), "tn_pdf_resample")
    return s, e


def weights_resample(e_bins_prev: Tensor, density_prev: Tensor, s_bins_prev: Tensor, S: int, anneal: float, nears: Tensor, fars: Tensor,
                     jitter: Optional[Tensor] = None, want_median: bool = True):
    """weights_fwd of the previous level + pdf_resample into the next one in a single launch (tn_weights_resample).
    -> weights_prev [N,Sp], median_prev [N,1] or None, s_bins [N,S+1], e_bins [N,S+1]"""
    N, Sp = density_prev.shape
    dev = density_prev.device
    w = torch.empty((N, Sp), device=dev)
    med = torch.empty((N, 1), device=dev) if want_median else None
    s = torch.empty((N, S + 1), device=dev)
    e = torch.empty((N, S + 1), device=dev)
    check(_lib.load().tn_weights_resample(_f32(e_bins_prev, "e_bins_prev", (N, Sp + 1)), _f32(density_prev, "density_prev", (N, Sp)),
                                          _f32(s_bins_prev, "s_bins_prev", (N, Sp + 1)), Sp, float(anneal), _f32(_lin_table("pdf", S, dev), "u"),
                                          _ray_scalar(jitter, "jitter", N, True), _ray_scalar(nears, "nears", N), _ray_scalar(fars, "fars", N), N, S,
                                          _f32(w, "w"), _f32(med, "median", optional=True), _f32(s, "s"), _f32(e, "e"), _stream()), "tn_weights_resample")
    return w, med, s, e


def weights_fwd(e_bins: Tensor, density: Tensor, want_median: bool = False):
    N, S = density.shape
    w = torch.empty((N, S), device=density.device)
    med = torch.empty((N, 1), device=density.device) if want_median else None
    check(_lib.load().tn_weights_fwd(_f32(e_bins, "e_bins", (N, S + 1)), _f32(density, "density", (N, S)), N, S, _f32(w, "w"),
                                     _f32(med, "median", optional=True), _stream()), "tn_weights_fwd")
    return w, med


def weights_bwd(e_bins: Tensor, density: Tensor, weights: Tensor, d_weights: Tensor) -> Tensor:
    N, S = density.shape
    dd = torch.empty((N, S), device=density.device)
    check(_lib.load().tn_weights_bwd(_f32(e_bins, "e_bins", (N, S + 1)), _f32(density, "density", (N, S)), _f32(weights, "weights", (N, S)),
                                     _f32(d_weights, "d_weights", (N, S)), N, S, _f32(dd, "dd"), _stream()), "tn_weights_bwd")
    return dd


# ------------------------------------------------------------------------------------------------ proposal nets
def prop_density_fwd(net: PropNetParams, origins: Tensor, directions: Tensor, e_bins: Tensor) -> Tensor:
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    out = torch.empty((N, S), device=origins.device)
    s = net.cstruct()
    check(_lib.load().tn_prop_density_fwd(C.byref(s), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)),
                                          _f32(e_bins, "e_bins", (N, S + 1)), N, S, _f32(out, "density"), _stream()), "tn_prop_density_fwd")
    return out


def _nbytes(t: Optional[Tensor]) -> int:
    """size of a workspace tensor in bytes (the workspace_bytes argument behind every workspace pointer of the C ABI)"""
    return 0 if t is None else t.numel() * t.element_size()


_PROP_WS: dict = {}


def prop_density_bwd(net: PropNetParams, origins: Tensor, directions: Tensor, e_bins: Tensor, d_density: Tensor,
                     d_origins: Optional[Tensor] = None, d_directions: Optional[Tensor] = None, tag: str = "") -> None:
    """`tag` names the scratch buffer: calls that may be in flight at the same time (different streams) must use different tags."""
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    need = int(_lib.load().tn_prop_workspace_bytes(N * S))
    key = (str(origins.device), tag)
    ws = _PROP_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=origins.device)
        _PROP_WS[key] = ws
    s = net.cstruct(need_grad=True)
    check(_lib.load().tn_prop_density_bwd(C.byref(s), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)),
                                          _f32(e_bins, "e_bins", (N, S + 1)), _f32(d_density, "d_density", (N, S)), N, S, C.c_void_p(ws.data_ptr()), _nbytes(ws),
                                          _f32(d_origins, "d_origins", (N, 3), True), _f32(d_directions, "d_directions", (N, 3), True), _stream()),
          "tn_prop_density_bwd")


# ------------------------------------------------------------------------------------------------ main field
def field_pack(fld: FieldParams, num_points: int, training: bool, tag: str = "main") -> Tensor:
    ws = fld.workspace(num_points, training, tag)
    s = fld.cstruct()
    check(_lib.load().tn_field_pack_weights(C.byref(s), C.c_void_p(ws.data_ptr()), _stream()), "tn_field_pack_weights")
    return ws


def field_fwd(fld: FieldParams, origins: Tensor, directions: Tensor, cam: Tensor, e_bins: Tensor, training: bool, want_pre: bool = False,
              tag: str = "main"):
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    ws = field_pack(fld, N * S, training, tag)
    dens = torch.empty((N, S), device=origins.device)
    rgb = torch.empty((N, S, fld.num_channels), device=origins.device)
    pre = torch.empty((N, S), device=origins.device) if want_pre else None
    s = fld.cstruct()
    check(_lib.load().tn_field_fwd(C.byref(s), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)), _i64(cam, "camera_indices", (N,)),
                                   _f32(e_bins, "e_bins", (N, S + 1)), N, S, 1 if training else 0, C.c_void_p(ws.data_ptr()), _nbytes(ws), _f32(dens, "density"),
                                   _f32(rgb, "rgb"), _f32(pre, "pre", optional=True), _stream()), "tn_field_fwd")
    return dens, rgb, pre


def field_density_fwd(fld: FieldParams, origins: Tensor, directions: Tensor, e_bins: Tensor, repack: bool = True, training: bool = False,
                      tag: str = "density_only") -> Tensor:
    """get_density alone.  training=True keeps the activations of the density path in the workspace `tag` for field_bwd(..., d_rgb=None)."""
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    # its own scratch (by tag) so that the workspace of the full forward (saved activations) is not clobbered
    ws = fld.workspace(N * S, training, tag)
    s = fld.cstruct()
    if repack:
        check(_lib.load().tn_field_pack_weights(C.byref(s), C.c_void_p(ws.data_ptr()), _stream()), "tn_field_pack_weights")
    dens = torch.empty((N, S), device=origins.device)
    check(_lib.load().tn_field_density_fwd(C.byref(s), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)),
                                           _f32(e_bins, "e_bins", (N, S + 1)), N, S, 1 if training else 0, C.c_void_p(ws.data_ptr()), _nbytes(ws),
                                           _f32(dens, "density"), _stream()), "tn_field_density_fwd")
    return dens


def field_bwd(fld: FieldParams, origins: Tensor, directions: Tensor, cam: Tensor, e_bins: Tensor, d_density: Tensor, d_rgb: Optional[Tensor],
              d_origins: Optional[Tensor] = None, d_directions: Optional[Tensor] = None, tag: str = "main") -> None:
    """d_rgb=None: density-only backward of field_density_fwd(training=True, tag=tag)."""
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    ws = fld.workspace(N * S, True, tag)
    s = fld.cstruct(need_grad=True)
    check(_lib.load().tn_field_bwd(C.byref(s), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)), _i64(cam, "camera_indices", (N,)),
                                   _f32(e_bins, "e_bins", (N, S + 1)), _f32(d_density, "d_density", (N, S)), _f32(d_rgb, "d_rgb", (N, S, fld.num_channels), True),
                                   N, S, C.c_void_p(ws.data_ptr()), _nbytes(ws), _f32(d_origins, "d_origins", (N, 3), True),
                                   _f32(d_directions, "d_directions", (N, 3), True), _stream()), "tn_field_bwd")


def field_bwd_phase(fld: FieldParams, origins: Tensor, directions: Tensor, cam: Tensor, e_bins: Tensor, d_density: Tensor, d_rgb: Tensor,
                    d_origins: Optional[Tensor], d_directions: Optional[Tensor], phases: int, level_begin: int = 0, level_end: int = 0,
                    tag: str = "main") -> None:
    """tn_field_bwd in phases (ops._lib.TN_BWD_MLP | TN_BWD_SCATTER | TN_BWD_JOIN): the table gradient of levels [level_begin, level_end) is final
    when its scatter has run, so a data-parallel caller can all-reduce it while the next level range is scattered."""
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    ws = fld.workspace(N * S, True, tag)
    s = fld.cstruct(need_grad=True)
    check(_lib.load().tn_field_bwd_phase(C.byref(s), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)),
                                         _i64(cam, "camera_indices", (N,)), _f32(e_bins, "e_bins", (N, S + 1)), _f32(d_density, "d_density", (N, S)),
                                         _f32(d_rgb, "d_rgb", (N, S, fld.num_channels)), N, S, C.c_void_p(ws.data_ptr()), _nbytes(ws),
                                         _f32(d_origins, "d_origins", (N, 3), True), _f32(d_directions, "d_directions", (N, 3), True),
                                         phases, level_begin, level_end, _stream()), "tn_field_bwd_phase")


def field_dense_count(fld: FieldParams, num_points: int, level_begin: int, level_end: int) -> int:
    """float2 cells of table levels [level_begin, level_end) if all of them are accumulated densely at this batch size, else 0."""
    s = fld.cstruct(need_grad=True)
    return int(_lib.load().tn_field_dense_count(C.byref(s), num_points, level_begin, level_end))


def field_bwd_scatter_dense(fld: FieldParams, origins: Tensor, directions: Tensor, e_bins: Tensor, d_origins: Optional[Tensor],
                            d_directions: Optional[Tensor], level_begin: int, level_end: int, dense_sum: Tensor, tag: str = "main") -> None:
    """The scatter phase of field_bwd_phase for a range of dense levels, with the per-cell sums written to dense_sum [n,2] instead of the table."""
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    ws = fld.workspace(N * S, True, tag)
    s = fld.cstruct(need_grad=True)
    check(_lib.load().tn_field_bwd_scatter_dense(C.byref(s), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)),
                                                 _f32(e_bins, "e_bins", (N, S + 1)), N, S, C.c_void_p(ws.data_ptr()), _nbytes(ws),
                                                 _f32(d_origins, "d_origins", (N, 3), True), _f32(d_directions, "d_directions", (N, 3), True),
                                                 level_begin, level_end, _f32(dense_sum, "dense_sum"), _stream()), "tn_field_bwd_scatter_dense")


def field_dense_fold(fld: FieldParams, num_points: int, level_begin: int, level_end: int, dense_sum: Tensor) -> None:
    """Hash the (exchanged) per-cell sums into the table gradient of levels [level_begin, level_end)."""
    s = fld.cstruct(need_grad=True)
    check(_lib.load().tn_field_dense_fold(C.byref(s), num_points, level_begin, level_end, _f32(dense_sum, "dense_sum"), _stream()), "tn_field_dense_fold")


_SCATTER_WS: dict = {}


def hash_scatter(table: Tensor, table_grad: Tensor, num_levels: int, log2_hashmap_size: int, res, origins: Tensor, directions: Tensor, e_bins: Tensor,
                 g_enc: Tensor, d_origins: Optional[Tensor] = None, d_directions: Optional[Tensor] = None, use_workspace: bool = True,
                 grad_is_zero: bool = False) -> None:
    """Backward of the hash encoding wrt the table (+ positions): trilinear scatter-add of g_enc [N*S, ld] -- or, level-major, [num_levels, N*S, 2]
    -- into table_grad.  use_workspace=False adds every level straight into the hashed gradient (no dense replicas for the coarse levels).
    grad_is_zero: the promise of TnGrid.table_grad_is_zero (table_grad holds zeros now: the fold stores instead of adding)."""
    N, S = e_bins.shape[0], e_bins.shape[1] - 1
    level_major = g_enc.dim() == 3
    if level_major and tuple(g_enc.shape) != (num_levels, N * S, 2):
        raise ValueError(f"level-major g_enc must be [{num_levels}, {N * S}, 2], got {tuple(g_enc.shape)}")
    ld = -1 if level_major else g_enc.shape[1]
    ws = None
    if use_workspace:
        need = int(_lib.load().tn_hash_scatter_workspace_bytes(N * S, num_levels))
        ws = _SCATTER_WS.get(str(origins.device))
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=origins.device)
            _SCATTER_WS[str(origins.device)] = ws
    g = _grid_struct(table, table_grad, num_levels, log2_hashmap_size, res, grad_is_zero=grad_is_zero)
    check(_lib.load().tn_hash_scatter(C.byref(g), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)), _f32(e_bins, "e_bins", (N, S + 1)),
                                      _f32(g_enc, "g_enc", None if level_major else (N * S, ld)), ld, N, S, _f32(d_origins, "d_origins", (N, 3), True),
                                      _f32(d_directions, "d_directions", (N, 3), True), C.c_void_p(ws.data_ptr()) if ws is not None else None, _nbytes(ws),
                                      _stream()), "tn_hash_scatter")


# ------------------------------------------------------------------------------------------------ renderers
def composite_fwd(rgb: Tensor, weights: Tensor, e_bins: Tensor, training: bool, want_depth: bool = True):
    """-> comp [N,C], accumulation [N,1], depth_median [N,1], depth_expected [N,1] (clipped to the batch-global midpoint range)."""
    N, S, Cc = rgb.shape
    dev = rgb.device
    comp = torch.empty((N, Cc), device=dev)
    acc = torch.empty((N, 1), device=dev)
    med = torch.empty((N, 1), device=dev) if want_depth else None
    exp = torch.empty((N, 1), device=dev) if want_depth else None
    mm = torch.empty(2, dtype=torch.int32, device=dev) if want_depth else None
    lib = _lib.load()
    if want_depth:
        check(lib.tn_minmax_init(C.c_void_p(mm.data_ptr()), _stream()), "tn_minmax_init")
    check(lib.tn_composite_fwd(_f32(rgb, "rgb", (N, S, Cc)), _f32(weights, "weights", (N, S)), _f32(e_bins, "e_bins", (N, S + 1)), N, S, Cc,
                               1 if training else 0, _f32(comp, "comp"), _f32(acc, "acc"), _f32(med, "med", optional=True), _f32(exp, "exp", optional=True),
                               C.c_void_p(mm.data_ptr()) if want_depth else None, _stream()), "tn_composite_fwd")
    if want_depth:
        check(lib.tn_clip_depth(_f32(exp, "exp"), C.c_void_p(mm.data_ptr()), N, _stream()), "tn_clip_depth")
    return comp, acc, med, exp


def composite_bwd(rgb: Tensor, weights: Tensor, d_comp: Tensor, d_weights: Tensor) -> Tensor:
    N, S, Cc = rgb.shape
    d_rgb = torch.empty_like(rgb)
    check(_lib.load().tn_composite_bwd(_f32(rgb, "rgb", (N, S, Cc)), _f32(weights, "weights", (N, S)), _f32(d_comp, "d_comp", (N, Cc)), N, S, Cc,
                                       _f32(d_rgb, "d_rgb"), _f32(d_weights, "d_weights", (N, S)), _stream()), "tn_composite_bwd")
    return d_rgb


def render_rays_eval(props: Sequence[PropNetParams], fld: FieldParams, origins: Tensor, directions: Tensor, cam: Tensor, nears: Tensor, fars: Tensor,
                     counts: Sequence[int], anneal: float):
    """The inference render of one branch in ONE library call (tn_render_rays_eval): proposal sampling, field, weights, renderers.
    -> dict(rgb [N,C], accumulation [N,1], depth [N,1], expected_depth [N,1], prop_depth_0/1 [N,1], density [N,S2], e_bins [N,S2+1],
    rgb_samples [N,S2,C])."""
    N = origins.shape[0]
    S0, S1, S2 = (int(c) for c in counts)
    Cc = fld.num_channels
    dev = origins.device
    lib = _lib.load()
    need = int(lib.tn_render_rays_eval_workspace_bytes(N, S0, S1, S2, Cc))
    if need < 0:
        raise ValueError("render_rays_eval: unsupported sample counts / channels")
    # a fresh buffer per call (the caching allocator recycles it): the per-level tensors returned below are views of it
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out = {"rgb": torch.empty((N, Cc), device=dev), "accumulation": torch.empty((N, 1), device=dev), "depth": torch.empty((N, 1), device=dev),
           "expected_depth": torch.empty((N, 1), device=dev), "prop_depth_0": torch.empty((N, 1), device=dev),
           "prop_depth_1": torch.empty((N, 1), device=dev), "density": torch.empty((N, S2), device=dev),
           "e_bins": torch.empty((N, S2 + 1), device=dev), "rgb_samples": torch.empty((N, S2, Cc), device=dev)}
    p0, p1, f = props[0].cstruct(), props[1].cstruct(), fld.cstruct()
    check(lib.tn_render_rays_eval(C.byref(p0), C.byref(p1), C.byref(f), _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)),
                                  _i64(cam, "camera_indices", (N,)), _ray_scalar(nears, "nears", N), _ray_scalar(fars, "fars", N), N, S0, S1, S2,
                                  float(anneal), _f32(_lin_table("spaced", S0, dev), "lin"), _f32(_lin_table("pdf", S1, dev), "u1"),
                                  _f32(_lin_table("pdf", S2, dev), "u2"), C.c_void_p(ws.data_ptr()), _nbytes(ws), _f32(out["rgb"], "rgb"), _f32(out["accumulation"], "acc"),
                                  _f32(out["depth"], "depth"), _f32(out["expected_depth"], "exp"), _f32(out["prop_depth_0"], "pd0"),
                                  _f32(out["prop_depth_1"], "pd1"), _f32(out["density"], "density"), _f32(out["e_bins"], "e_bins"),
                                  _f32(out["rgb_samples"], "rgb_samples"), _stream()), "tn_render_rays_eval")
    # the sampler's intermediate levels live in the workspace, in the order tn_render_rays_eval lays them out (csrc/tn_pipeline.hip,
    # eval_layout: every region rounded up to 256 bytes): s_bins, e_bins, density, weights of level 0, the same of level 1, s_bins of level 2,
    # (its e_bins: returned above), weights of level 2
    off = 0

    def view(rows, cols):
        nonlocal off
        n = rows * cols
        t = ws[off:off + 4 * n].view(torch.float32).view(rows, cols)
        off += (4 * n + 255) // 256 * 256
        return t

    lv = []
    for S in (S0, S1):
        lv.append({"s_bins": view(N, S + 1), "e_bins": view(N, S + 1), "density": view(N, S), "weights": view(N, S)})
    s2 = view(N, S2 + 1)
    view(N, S2 + 1)  # e_bins of level 2 inside the workspace: unused, the caller's tensor received them
    lv.append({"s_bins": s2, "e_bins": out["e_bins"], "density": out["density"], "weights": view(N, S2)})
    out["levels"] = lv
    return out


_TRAIN_LAYOUTS: dict = {}


def render_rays_train(props: Sequence[PropNetParams], fld: FieldParams, pose: Optional[Tensor], frozen: Optional[Tensor], origins: Tensor,
                      directions: Tensor, cam: Tensor, nears: Tensor, fars: Tensor, counts: Sequence[int], anneal: float,
                      jitters: Optional[Sequence[Optional[Tensor]]] = None, tag: str = "main", wait_event=None, zero_fill: Optional[Tensor] = None,
                      save_prop_enc: bool = False):
    """The training forward of one branch in ONE library call (tn_render_rays_train): pose correction, proposal sampling with jitter, field
    (activations kept in the field's workspace `tag`), weights, renderers.  Every result is a view of one allocation.
    wait_event: a torch.cuda.Event the stream waits for right before the field's first parameter read (the previous step's Adam, see engine).
    zero_fill: a float tensor (whole multiples of 4 elements, 16-byte aligned) the call clears inside the field's first launch.
    save_prop_enc: the proposal networks take a gradient this iteration: their encodings are kept in the buffer for render_rays_train_bwd.
    -> dict(origins, directions [N,3]; levels: 3 x dict(s_bins, e_bins, density, weights, median); rgb_samples [N,S2,C]; rgb [N,C];
    accumulation, depth, expected_depth [N,1])."""
    N = origins.shape[0]
    S0, S1, S2 = (int(c) for c in counts)
    Cc = fld.num_channels
    dev = origins.device
    lib = _lib.load()
    key = (N, S0, S1, S2, Cc)
    off = _TRAIN_LAYOUTS.get(key)
    if off is None:
        arr = (C.c_int64 * _lib.TN_RENDER_TRAIN_OFFSETS)()
        check(lib.tn_render_rays_train_layout(N, S0, S1, S2, Cc, arr, _lib.TN_RENDER_TRAIN_OFFSETS), "tn_render_rays_train_layout")
        off = _TRAIN_LAYOUTS[key] = [int(v) for v in arr]
    buf = torch.empty(off[_lib.TN_RENDER_TRAIN_OFFSETS - 1], device=dev)
    ws = fld.workspace(N * S2, True, tag)
    jit = list(jitters) if jitters is not None else [None, None, None]
    Cn = pose.shape[0] if pose is not None else 0
    p0, p1, f = props[0].cstruct(), props[1].cstruct(), fld.cstruct()
    check(lib.tn_render_rays_train(C.byref(p0), C.byref(p1), C.byref(f), _f32(pose, "pose", (Cn, 6), optional=True), _u8(frozen, Cn) if pose is not None else None,
                                   Cn, _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3)), _i64(cam, "camera_indices", (N,)),
                                   _ray_scalar(nears, "nears", N), _ray_scalar(fars, "fars", N), N, S0, S1, S2, float(anneal),
                                   _ray_scalar(jit[0], "jitter", N, True), _ray_scalar(jit[1], "jitter", N, True), _ray_scalar(jit[2], "jitter", N, True),
                                   _f32(_lin_table("spaced", S0, dev), "lin"), _f32(_lin_table("pdf", S1, dev), "u1"), _f32(_lin_table("pdf", S2, dev), "u2"),
                                   C.c_void_p(ws.data_ptr()), _nbytes(ws), C.c_void_p(buf.data_ptr()),
                                   C.c_void_p(wait_event.cuda_event) if wait_event is not None else None,
                                   C.c_void_p(zero_fill.data_ptr()) if zero_fill is not None else None, _nbytes(zero_fill), 1 if save_prop_enc else 0,
                                   _stream()), "tn_render_rays_train")

    so = buf.storage_offset()

    def v(slot, *shape):  # one as_strided per result (a slice + a view are two dispatcher calls, and a step hands out 22 of these)
        return buf.as_strided(shape, _contig_strides(shape), so + off[slot])

    levels = []
    for i, S in enumerate((S0, S1)):
        b = 2 + 5 * i
        levels.append({"s_bins": v(b, N, S + 1), "e_bins": v(b + 1, N, S + 1), "density": v(b + 2, N, S), "weights": v(b + 3, N, S), "median": v(b + 4, N, 1)})
    levels.append({"s_bins": v(12, N, S2 + 1), "e_bins": v(13, N, S2 + 1), "density": v(14, N, S2), "weights": v(15, N, S2), "median": v(19, N, 1)})
    return {"buf": buf, "prop_enc_saved": bool(save_prop_enc), "origins": v(0, N, 3) if pose is not None else origins, "directions": v(1, N, 3) if pose is not None else directions, "levels": levels,
            "rgb_samples": v(16, N, S2, Cc), "rgb": v(17, N, Cc), "accumulation": v(18, N, 1), "depth": v(19, N, 1), "expected_depth": v(20, N, 1)}


class TrainStepCall:
    """One training iteration of the shared-density model as ONE library call (tn_train_step): the argument block is kept between iterations,
    only what changes per iteration is written again.  RenderEngine.train_step builds one per (engine, scaler) and calls run() every step; the
    five calls it replaces stay the reference for what it enqueues (tests/test_hip_ops_gpu.py)."""

    def __init__(self, props: Sequence[PropNetParams], fld: FieldParams, pose: Tensor, frozen: Optional[Tensor], pose_grad: Tensor, penalties, counts,
                 mults, arena_tensors, check_ranges, pose_flag: int, scaler, tags=("main", "side0", "side1")):
        """penalties = (trans, rot, scale); mults = (thermal, tv, cross, distortion, interlevel); arena_tensors = (params, grads, exp_avg,
        exp_avg_sq); check_ranges = [(lo, hi, flag)] small gradient ranges for GradScaler's check; scaler: optim.DeviceGradScaler."""
        self.props, self.fld, self.pose, self.frozen, self.pose_grad = list(props), fld, pose, frozen, pose_grad
        self.counts = tuple(int(c) for c in counts)
        self.tags = tags
        self.scaler = scaler
        self.arena_tensors = arena_tensors
        if fld.num_channels != 4 or len(check_ranges) > _lib.TN_TRAIN_STEP_MAX_RANGES:
            raise ValueError("tn_train_step is the shared-density iteration (4 channels) with at most 8 small gradient ranges")
        st = self.st = _lib.TnTrainStep()
        Cn = pose.shape[0]
        st.pose_adjustment, st.frozen, st.num_cameras = _f32(pose, "pose", (Cn, 6)), _u8(frozen, Cn), Cn
        st.grad_pose = _f32(pose_grad, "grad_pose", (Cn, 6))
        st.trans_pen, st.rot_pen, st.pen_scale = (float(x) for x in penalties)
        st.S0, st.S1, st.S2 = self.counts
        st.thermal_mult, st.tv_mult, st.cross_mult, st.distortion_mult, st.interlevel_mult = (float(x) for x in mults)
        st.num_check = len(check_ranges)
        for k, (lo, hi, fl) in enumerate(check_ranges):
            st.check_offsets[k], st.check_counts[k], st.check_flags[k] = int(lo), int(hi - lo), int(fl)
        st.pose_flag = int(pose_flag)
        params, grads, m, v = arena_tensors
        st.params, st.grads, st.exp_avg, st.exp_avg_sq = _f32(params, "params"), _f32(grads, "grads"), _f32(m, "exp_avg"), _f32(v, "exp_avg_sq")
        st.beta1, st.beta2, st.eps = 0.9, 0.999, 1e-15
        st.found_inf, st.num_flags = _f32(scaler.found_inf, "found_inf"), int(scaler.found_inf.numel())
        st.skipped, st.lag_index = C.c_void_p(scaler.skipped.data_ptr()), int(scaler.lag_index)
        sc, gt, done, gf, bf, gi = scaler.fused_update_args()
        st.scale, st.growth_tracker, st.done_counter = _f32(sc, "scale", (1,)), C.c_void_p(gt.data_ptr()), C.c_void_p(done.data_ptr())
        st.growth_factor, st.backoff_factor, st.growth_interval = float(gf), float(bf), int(gi)
        self._N = -1
        self._keep = None

    def _for_batch(self, N: int, dev) -> None:
        """what depends on the batch size only: layouts, workspaces, sampler tables"""
        st, (S0, S1, S2) = self.st, self.counts
        lib = _lib.load()
        key = (N, S0, S1, S2, 4)
        off = _TRAIN_LAYOUTS.get(key)
        if off is None:
            arr = (C.c_int64 * _lib.TN_RENDER_TRAIN_OFFSETS)()
            check(lib.tn_render_rays_train_layout(N, S0, S1, S2, 4, arr, _lib.TN_RENDER_TRAIN_OFFSETS), "tn_render_rays_train_layout")
            off = _TRAIN_LAYOUTS[key] = [int(x) for x in arr]
        self.off = off
        ws = self.fld.workspace(N * S2, True, self.tags[0])
        w0, w1 = _prop_ws(dev, N * S0, self.tags[1]), _prop_ws(dev, N * S1, self.tags[2])
        need = int(lib.tn_render_rays_train_bwd_tmp_floats(N, S0, S1, S2, 4))
        tkey = (str(dev), self.tags[0])
        tmp = _BWD_TMP.get(tkey)
        if tmp is None or tmp.numel() < need:
            tmp = _BWD_TMP[tkey] = torch.empty(need, device=dev)
        lins = (_lin_table("spaced", S0, dev), _lin_table("pdf", S1, dev), _lin_table("pdf", S2, dev))
        st.N = N
        st.field_workspace, st.field_workspace_bytes = C.c_void_p(ws.data_ptr()), _nbytes(ws)
        st.prop_workspace0, st.prop_workspace_bytes0 = C.c_void_p(w0.data_ptr()), _nbytes(w0)
        st.prop_workspace1, st.prop_workspace_bytes1 = C.c_void_p(w1.data_ptr()), _nbytes(w1)
        st.bwd_tmp = C.c_void_p(tmp.data_ptr())
        st.lin_spaced0, st.lin_pdf1, st.lin_pdf2 = (_f32(t, "lin") for t in lins)
        self._batch_keep = (ws, w0, w1, tmp, lins)
        self._N = N

    def run(self, origins: Tensor, directions: Tensor, cam: Tensor, image: Tensor, is_thermal: Tensor, nears: Tensor, fars: Tensor, anneal: float,
            jitters: Sequence[Tensor], prop_grad: bool, acc_flat: Tensor, acc: dict, ranges, sched_step: int, fwd_buf: Optional[Tensor] = None,
            next_plan: Optional[tuple] = None) -> Tensor:
        """acc: name -> view of acc_flat for L [16], Lp [LOSS_LINES,16], d_comp, dw0, dw1 (prop_grad only), dw2, d_o, d_d.
        ranges: [(lo, hi, adam step, lr_init, lr_final, max_steps, flag)] of the groups stepped this iteration.  -> the forward's buffer.
        fwd_buf: the buffer the PREVIOUS call's next_plan filled up to the field's bins for this very batch (TnTrainStep.sampling_done).
        next_plan = (jitters, anneal, prop_grad) of the NEXT iteration: its sampling front for the pending batch (sample_rays_deferred) runs as co-work of
        this call's optimiser launch (TnTrainStep.next_sampling); self.next_buf is the buffer it filled, or None when the library did not take it."""
        N = origins.shape[0]
        dev = origins.device
        if N != self._N:
            self._for_batch(N, dev)
        st, (S0, S1, S2) = self.st, self.counts
        # the structs' addresses must be the ones the library reads during THIS call: rebuilt only when a tensor of the networks moved
        p0, p1, f = self.props[0].cstruct(need_grad=True), self.props[1].cstruct(need_grad=True), self.fld.cstruct(need_grad=True)
        st.prop0, st.prop1, st.field = C.pointer(p0), C.pointer(p1), C.pointer(f)
        st.origins_in, st.directions_in = _f32(origins, "origins", (N, 3)), _f32(directions, "directions", (N, 3))
        st.camera_indices = _i64(cam, "camera_indices", (N,))
        st.image, st.is_thermal = _f32(image, "image", (N, 3)), _f32(is_thermal, "is_thermal", (N,))
        st.nears, st.fars = _ray_scalar(nears, "nears", N), _ray_scalar(fars, "fars", N)
        st.anneal, st.prop_grad = float(anneal), 1 if prop_grad else 0
        st.jitter0, st.jitter1, st.jitter2 = (_ray_scalar(j, "jitter", N) for j in jitters)
        total = self.off[_lib.TN_RENDER_TRAIN_OFFSETS - 1]
        if fwd_buf is not None and (fwd_buf.numel() != total or fwd_buf.device != dev):
            raise ValueError("fwd_buf is not this batch's forward buffer")
        buf = torch.empty(total, device=dev) if fwd_buf is None else fwd_buf
        st.fwd_out, st.sampling_done = C.c_void_p(buf.data_ptr()), 0 if fwd_buf is None else 1
        st.acc, st.acc_bytes = C.c_void_p(acc_flat.data_ptr()), _nbytes(acc_flat)
        st.losses16, st.loss_lines = _f32(acc["L"], "losses16", (16,)), _f32(acc["Lp"], "loss_lines", (LOSS_LINES, 16))
        st.d_comp, st.d_weights2 = _f32(acc["d_comp"], "d_comp", (N, 4)), _f32(acc["dw2"], "d_weights2", (N, S2))
        st.d_weights0 = _f32(acc.get("dw0"), "d_weights0", (N, S0), optional=True) if prop_grad else None
        st.d_weights1 = _f32(acc.get("dw1"), "d_weights1", (N, S1), optional=True) if prop_grad else None
        st.d_origins, st.d_directions = _f32(acc["d_o"], "d_origins", (N, 3)), _f32(acc["d_d"], "d_directions", (N, 3))
        if len(ranges) > _lib.TN_TRAIN_STEP_MAX_RANGES:
            raise ValueError("at most 8 Adam ranges per iteration")
        st.num_ranges = len(ranges)
        for k, (lo, hi, step, lr0, lr1, ms, fl) in enumerate(ranges):
            st.offsets[k], st.counts[k], st.steps[k] = int(lo), int(hi - lo), int(step)
            st.lrs[k], st.lr_finals[k], st.sched_max_steps[k], st.flag_index[k] = float(lr0), float(lr1), int(ms), int(fl)
        st.sched_step = int(sched_step)
        # GradScaler's growth / backoff settings are read from the scaler EVERY call: DeviceGradScaler.load_state_dict() on the same object (a resumed
        # run) changes them, and the five-call path reads fused_update_args() per step too -- the two paths must keep agreeing after a resume.
        _, _, _, gf, bf, gi = self.scaler.fused_update_args()
        st.growth_factor, st.backoff_factor, st.growth_interval = float(gf), float(bf), int(gi)
        # the next iteration's batch, when the data manager has handed one over (sample_rays_deferred): sampled inside this call's optimiser launch
        global _PENDING_SAMPLE
        pend = _PENDING_SAMPLE
        taken = C.c_int32(0)
        if pend is not None and len(ranges) > 0:
            st.next_sample, st.next_sample_taken = C.pointer(pend[0]), C.pointer(taken)
        else:
            st.next_sample, st.next_sample_taken = C.POINTER(_lib.TnSampleRays)(), C.POINTER(C.c_int32)()
        # the next iteration's sampling front for that batch, in the same co-work blocks
        ns_taken, nxt, next_buf = C.c_int32(0), None, None
        self.next_buf = None
        if next_plan is not None and pend is not None and len(ranges) > 0 and int(pend[0].num_rays) == N:
            njit, nanneal, nprop = next_plan
            next_buf = torch.empty(total, device=dev)
            nxt = _lib.TnNextSampling()
            nxt.fwd_out = C.c_void_p(next_buf.data_ptr())
            nxt.jitter0, nxt.jitter1, nxt.jitter2 = (_ray_scalar(j, "jitter", N) for j in njit)
            nxt.anneal, nxt.prop_grad = float(nanneal), 1 if nprop else 0
            st.next_sampling, st.next_sampling_taken = C.pointer(nxt), C.pointer(ns_taken)
        else:
            st.next_sampling, st.next_sampling_taken = C.POINTER(_lib.TnNextSampling)(), C.POINTER(C.c_int32)()
        self._keep = (p0, p1, f, buf, acc_flat, origins, directions, cam, image, is_thermal, jitters, pend, nxt, next_buf, next_plan)  # alive until the next call replaces them
        check(_lib.load().tn_train_step(C.byref(st), _stream()), "tn_train_step")
        if taken.value:
            _PENDING_SAMPLE = None  # (else it stays pending: the data manager launches it itself before it hands the batch out)
        if ns_taken.value:
            self.next_buf = next_buf
        return buf


def _prop_ws(device, num_points: int, tag: str) -> Tensor:
    need = int(_lib.load().tn_prop_workspace_bytes(num_points))
    key = (str(device), tag)
    ws = _PROP_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _PROP_WS[key] = ws
    return ws


_BWD_TMP: dict = {}


def render_rays_train_bwd(props: Sequence[PropNetParams], fld: FieldParams, fwd_buf: Tensor, origins: Tensor, directions: Tensor, cam: Tensor,
                          counts: Sequence[int], d_comp: Tensor, d_weights: Sequence[Optional[Tensor]], d_density_extra: Optional[Tensor],
                          d_origins: Optional[Tensor], d_directions: Optional[Tensor], tag: str = "main", side_tags=("side0", "side1"),
                          prop_enc_saved: bool = False) -> None:
    """The training backward of one branch in ONE library call (tn_render_rays_train_bwd): renderer backward, field backward (+ d position, table
    scatter) and -- when d_weights[0] / d_weights[1] are given -- both proposal networks' backward on the library's companion streams.
    fwd_buf: the buffer ops.render_rays_train returned ("buf"); origins / directions: its pose-corrected rays; d_weights = [level0, level1, fine]."""
    N = origins.shape[0]
    S0, S1, S2 = (int(c) for c in counts)
    Cc = fld.num_channels
    lib = _lib.load()
    dev = origins.device
    need = int(lib.tn_render_rays_train_bwd_tmp_floats(N, S0, S1, S2, Cc))
    key = (str(dev), tag)
    tmp = _BWD_TMP.get(key)
    if tmp is None or tmp.numel() < need:
        tmp = _BWD_TMP[key] = torch.empty(need, device=dev)
    ws = fld.workspace(N * S2, True, tag)
    prop_grad = d_weights[0] is not None
    p0 = props[0].cstruct(need_grad=True) if prop_grad else None
    p1 = props[1].cstruct(need_grad=True) if prop_grad else None
    f = fld.cstruct(need_grad=True)
    w0 = _prop_ws(dev, N * S0, side_tags[0]) if prop_grad else None
    w1 = _prop_ws(dev, N * S1, side_tags[1]) if prop_grad else None
    check(lib.tn_render_rays_train_bwd(C.byref(p0) if prop_grad else None, C.byref(p1) if prop_grad else None, C.byref(f), _f32(origins, "origins", (N, 3)),
                                       _f32(directions, "directions", (N, 3)), _i64(cam, "camera_indices", (N,)), N, S0, S1, S2, C.c_void_p(fwd_buf.data_ptr()),
                                       _f32(d_comp, "d_comp", (N, Cc)), _f32(d_weights[0], "d_weights0", (N, S0), True), _f32(d_weights[1], "d_weights1", (N, S1), True),
                                       _f32(d_weights[2], "d_weights2", (N, S2)), _f32(d_density_extra, "d_density_extra", (N, S2), True),
                                       C.c_void_p(ws.data_ptr()), _nbytes(ws), C.c_void_p(w0.data_ptr()) if prop_grad else None, _nbytes(w0) if prop_grad else 0,
                                       C.c_void_p(w1.data_ptr()) if prop_grad else None, _nbytes(w1) if prop_grad else 0,
                                       C.c_void_p(tmp.data_ptr()), _f32(d_origins, "d_origins", (N, 3), True), _f32(d_directions, "d_directions", (N, 3), True),
                                       1 if (prop_enc_saved and prop_grad) else 0, _stream()), "tn_render_rays_train_bwd")


_RENDER_SCRATCH: dict = {}


def render_fwd(e_bins: Tensor, density: Tensor, rgb: Tensor, training: bool, want_depth: bool = True):
    """weights_fwd + composite_fwd (+ depth clip) of the last level in one launch (tn_render_fwd)
    -> weights [N,S], comp [N,C], accumulation [N,1], depth_median [N,1] or None, depth_expected [N,1] or None."""
    N, S, Cc = rgb.shape
    dev = rgb.device
    w = torch.empty((N, S), device=dev)
    comp = torch.empty((N, Cc), device=dev)
    acc = torch.empty((N, 1), device=dev)
    med = torch.empty((N, 1), device=dev) if want_depth else None
    exp = torch.empty((N, 1), device=dev) if want_depth else None
    st = _stream()
    scratch = None
    if want_depth:
        # per-block min / max of the sample midpoints (contents irrelevant between calls): one buffer per (device, stream)
        key = (dev.index, st.value)
        scratch = _RENDER_SCRATCH.get(key)
        if scratch is None:
            scratch = _RENDER_SCRATCH[key] = torch.empty(_lib.TN_RENDER_SCRATCH_FLOATS, device=dev)
    check(_lib.load().tn_render_fwd(_f32(e_bins, "e_bins", (N, S + 1)), _f32(density, "density", (N, S)), _f32(rgb, "rgb", (N, S, Cc)), N, S, Cc,
                                    1 if training else 0, _f32(w, "w"), _f32(comp, "comp"), _f32(acc, "acc"), _f32(med, "med", optional=True),
                                    _f32(exp, "exp", optional=True), _f32(scratch, "scratch", optional=True), st), "tn_render_fwd")
    return w, comp, acc, med, exp


def render_bwd(e_bins: Tensor, density: Tensor, rgb: Tensor, weights: Tensor, d_comp: Tensor, d_weights_in: Tensor):
    """composite_bwd + weights_bwd in one launch (tn_render_bwd) -> d_rgb [N,S,C], d_density [N,S]; d_weights_in is left as it is."""
    N, S, Cc = rgb.shape
    d_rgb = torch.empty_like(rgb)
    dd = torch.empty((N, S), device=rgb.device)
    check(_lib.load().tn_render_bwd(_f32(e_bins, "e_bins", (N, S + 1)), _f32(density, "density", (N, S)), _f32(rgb, "rgb", (N, S, Cc)),
                                    _f32(weights, "weights", (N, S)), _f32(d_comp, "d_comp", (N, Cc)), _f32(d_weights_in, "d_weights", (N, S)), N, S, Cc,
                                    _f32(d_rgb, "d_rgb"), _f32(dd, "d_density"), _stream()), "tn_render_bwd")
    return d_rgb, dd


# ------------------------------------------------------------------------------------------------ losses
def distortion_loss(s_bins: Tensor, weights: Tensor, mult: float, loss_out: Tensor, d_weights: Optional[Tensor]) -> None:
    N, S = weights.shape
    check(_lib.load().tn_distortion_loss(_f32(s_bins, "s_bins", (N, S + 1)), _f32(weights, "weights", (N, S)), N, S, float(mult), _f32(loss_out, "loss"),
                                         _f32(d_weights, "d_weights", (N, S), True), _stream()), "tn_distortion_loss")


def interlevel_loss(s_fine: Tensor, w_fine: Tensor, s_prop: Tensor, w_prop: Tensor, mult: float, loss_out: Tensor, d_w_prop: Optional[Tensor]) -> None:
    N, Sf = w_fine.shape
    Sp = w_prop.shape[1]
    check(_lib.load().tn_interlevel_loss(_f32(s_fine, "s_fine", (N, Sf + 1)), _f32(w_fine, "w_fine", (N, Sf)), Sf, _f32(s_prop, "s_prop", (N, Sp + 1)),
                                         _f32(w_prop, "w_prop", (N, Sp)), Sp, N, float(mult), _f32(loss_out, "loss"), _f32(d_w_prop, "d_w_prop", (N, Sp), True),
                                         _stream()), "tn_interlevel_loss")


def _prop_level_arrays(props, N: int):
    n = len(props)
    sb = (C.c_void_p * max(n, 1))()
    wp = (C.c_void_p * max(n, 1))()
    dw = (C.c_void_p * max(n, 1))()
    sp = (C.c_int32 * max(n, 1))()
    for i, (s_p, w_p, d_p) in enumerate(props):
        Sp = w_p.shape[1]
        sb[i] = _f32(s_p, "s_prop", (N, Sp + 1)).value
        wp[i] = _f32(w_p, "w_prop", (N, Sp)).value
        v = _f32(d_p, "d_w_prop", (N, Sp), True)
        dw[i] = v.value if v is not None else None
        sp[i] = Sp
    return n, sb, wp, sp, dw


def proposal_losses(s_fine: Tensor, w_fine: Tensor, props, distortion_mult: float, interlevel_mult: float, distortion_out: Tensor,
                    interlevel_out: Tensor, d_w_fine: Optional[Tensor]) -> None:
    """distortion_loss on the fine level + interlevel_loss against every proposal level, one launch (tn_proposal_losses).
    props: list of (s_bins [N,Sp+1], weights [N,Sp], d_weights [N,Sp] or None)."""
    N, Sf = w_fine.shape
    n, sb, wp, sp, dw = _prop_level_arrays(props, N)
    check(_lib.load().tn_proposal_losses(_f32(s_fine, "s_fine", (N, Sf + 1)), _f32(w_fine, "w_fine", (N, Sf)), Sf, n, sb, wp, sp, dw, N,
                                         float(distortion_mult), float(interlevel_mult), _f32(distortion_out, "distortion"),
                                         _f32(interlevel_out, "interlevel"), _f32(d_w_fine, "d_w_fine", (N, Sf), True), _stream()),
          "tn_proposal_losses")


LOSS_LINES = _lib.TN_LOSS_LINES


def train_losses(s_fine: Tensor, w_fine: Tensor, props, distortion_mult: float, interlevel_mult: float, d_w_fine: Optional[Tensor],
                 loss_lines: Tensor, pixel=None) -> None:
    """proposal_losses and (pixel != None: the arguments of pixel_losses() without losses_out) the pixel terms in ONE launch, the sums spread
    over loss_lines [LOSS_LINES,16] (zero-filled by the caller; slots: 0 rgb 1 thermal 2 tv 3 cross 4/5 ray counts 8 interlevel 9 distortion);
    losses_finish() adds the lines up (tn_train_losses / tn_losses_finish)."""
    N, Sf = w_fine.shape
    n, sb, wp, sp, dw = _prop_level_arrays(props, N)
    if pixel is None:
        tail = (None, 0, None, 0, None, None, 0.0, 0.0, 0.0, None, None)
    else:
        pr, pt, image, is_th, tm, tvm, cm, d_pr, d_pt = pixel
        a = _pixel_args(pr, pt, image, is_th, tm, tvm, cm, None, d_pr, d_pt, N=N)
        tail = a[:9] + a[10:]
    check(_lib.load().tn_train_losses(_f32(s_fine, "s_fine", (N, Sf + 1)), _f32(w_fine, "w_fine", (N, Sf)), Sf, n, sb, wp, sp, dw, N,
                                      float(distortion_mult), float(interlevel_mult), _f32(d_w_fine, "d_w_fine", (N, Sf), True), *tail,
                                      _f32(loss_lines, "loss_lines", (LOSS_LINES, 16)), _stream()), "tn_train_losses")


def render_losses_bwd(e_bins: Tensor, density: Tensor, rgb: Tensor, s_fine: Tensor, props, distortion_mult: float, interlevel_mult: float,
                      d_w_fine: Tensor, image: Tensor, is_thermal: Tensor, thermal_mult: float, tv_mult: float, cross_mult: float, d_comp: Tensor,
                      loss_lines: Tensor, clip_depth: bool = True):
    """render_fwd(training) + train_losses + render_bwd of the shared-density model's last level in ONE launch (tn_render_losses_bwd).
    rgb [N,S,4]; props as train_losses; d_w_fine [N,S], d_comp [N,4] and the props' d weights are accumulators.
    -> weights, comp [N,4], accumulation, depth_median, depth_expected, d_rgb [N,S,4], d_density [N,S]."""
    N, S, Cc = rgb.shape
    dev = rgb.device
    w = torch.empty((N, S), device=dev)
    comp = torch.empty((N, Cc), device=dev)
    acc = torch.empty((N, 1), device=dev)
    med = torch.empty((N, 1), device=dev)
    exp = torch.empty((N, 1), device=dev)
    d_rgb = torch.empty((N, S, Cc), device=dev)
    dd = torch.empty((N, S), device=dev)
    st = _stream()
    key = (dev.index, st.value)
    scratch = _RENDER_SCRATCH.get(key)
    if scratch is None:
        scratch = _RENDER_SCRATCH[key] = torch.empty(_lib.TN_RENDER_SCRATCH_FLOATS, device=dev)
    n, sb, wp, sp, dw = _prop_level_arrays(props, N)
    check(_lib.load().tn_render_losses_bwd(_f32(e_bins, "e_bins", (N, S + 1)), _f32(density, "density", (N, S)), _f32(rgb, "rgb", (N, S, Cc)), N, S, Cc,
                                           _f32(w, "w"), _f32(comp, "comp"), _f32(acc, "acc"), _f32(med, "med"), _f32(exp, "exp"), _f32(scratch, "scratch"),
                                           _f32(s_fine, "s_fine", (N, S + 1)), n, sb, wp, sp, dw, float(distortion_mult), float(interlevel_mult),
                                           _f32(d_w_fine, "d_w_fine", (N, S)), _f32(image, "image", (N, 3)), _f32(is_thermal, "is_thermal", (N,)),
                                           float(thermal_mult), float(tv_mult), float(cross_mult), _f32(d_comp, "d_comp", (N, Cc)),
                                           _f32(loss_lines, "loss_lines", (LOSS_LINES, 16)), _f32(d_rgb, "d_rgb"), _f32(dd, "d_density"),
                                           1 if clip_depth else 0, st), "tn_render_losses_bwd")
    return w, comp, acc, med, exp, d_rgb, dd


def losses_finish(loss_lines: Tensor, losses16: Tensor, pose: Optional[Tensor] = None, trans_pen: float = 0.0, rot_pen: float = 0.0,
                  scale: float = 0.0, reg_out: Optional[Tensor] = None, grad_pose: Optional[Tensor] = None) -> None:
    """losses16[k] += column sums of loss_lines; with `pose` also camera_reg(pose, ...) -> reg_out / grad_pose, same single-block launch."""
    if losses16.numel() < 16:
        raise ValueError("losses16 needs 16 floats")
    Cn = pose.shape[0] if pose is not None else 0
    check(_lib.load().tn_losses_finish(_f32(loss_lines, "loss_lines", (LOSS_LINES, 16)), _f32(losses16, "losses16"),
                                       _f32(pose, "pose", (Cn, 6), optional=True), Cn, float(trans_pen), float(rot_pen), float(scale),
                                       _f32(reg_out, "reg_out", optional=True), _f32(grad_pose, "grad_pose", (Cn, 6), True), _stream()),
          "tn_losses_finish")


def _pixel_args(pred_rgb: Tensor, pred_thermal: Tensor, image: Tensor, is_thermal: Tensor, thermal_mult: float, tv_mult: float, cross_mult: float,
                losses_out: Tensor, d_pred_rgb: Optional[Tensor], d_pred_thermal: Optional[Tensor], N: Optional[int] = None):
    """Validated argument tail shared by tn_pixel_losses and tn_train_losses (without N and the stream)."""
    n = pred_rgb.shape[0]
    if N is not None and n != N:
        raise ValueError("pixel terms and proposal terms must cover the same rays")
    if pred_rgb.dtype != torch.float32 or pred_thermal.dtype != torch.float32 or not pred_rgb.is_cuda:
        raise ValueError("predictions must be CUDA float32")
    if pred_rgb.stride(1) != 1 or pred_thermal.stride(1) != 1 or pred_rgb.shape != (n, 3) or pred_thermal.shape != (n, 1):
        raise ValueError("bad prediction views")
    if losses_out is not None and losses_out.numel() < 8:
        raise ValueError("losses_out needs 8 floats")
    for g, p in ((d_pred_rgb, pred_rgb), (d_pred_thermal, pred_thermal)):
        if g is not None and (g.stride() != p.stride() or g.shape != p.shape or g.dtype != torch.float32):
            raise ValueError("gradient views must mirror the prediction views")
    return (C.c_void_p(pred_rgb.data_ptr()), pred_rgb.stride(0), C.c_void_p(pred_thermal.data_ptr()), pred_thermal.stride(0),
            _f32(image, "image", (n, 3)), _f32(is_thermal, "is_thermal", (n,)), float(thermal_mult), float(tv_mult), float(cross_mult),
            _f32(losses_out, "losses", optional=True), C.c_void_p(d_pred_rgb.data_ptr()) if d_pred_rgb is not None else None,
            C.c_void_p(d_pred_thermal.data_ptr()) if d_pred_thermal is not None else None)


def pixel_losses(pred_rgb: Tensor, pred_thermal: Tensor, image: Tensor, is_thermal: Tensor, thermal_mult: float, tv_mult: float, cross_mult: float,
                 losses_out: Tensor, d_pred_rgb: Optional[Tensor], d_pred_thermal: Optional[Tensor]) -> None:
    """pred_rgb [N,3] / pred_thermal [N,1] may be strided views of one [N,4] buffer (shared mode)."""
    a = _pixel_args(pred_rgb, pred_thermal, image, is_thermal, thermal_mult, tv_mult, cross_mult, losses_out, d_pred_rgb, d_pred_thermal)
    # C order: ..., image, is_thermal, N, thermal_mult, ...
    check(_lib.load().tn_pixel_losses(*a[:6], pred_rgb.shape[0], *a[6:], _stream()), "tn_pixel_losses")


def l1_loss(x: Tensor, y: Tensor, gx: float, gy: float, loss_out: Tensor, d_x: Optional[Tensor], d_y: Optional[Tensor]) -> None:
    n = x.numel()
    check(_lib.load().tn_l1_loss(_f32(x, "x"), _f32(y, "y", tuple(x.shape)), n, float(gx), float(gy), _f32(loss_out, "loss"),
                                 _f32(d_x, "d_x", tuple(x.shape), True), _f32(d_y, "d_y", tuple(x.shape), True), _stream()), "tn_l1_loss")


def camera_reg(pose: Tensor, trans_pen: float, rot_pen: float, scale: float, loss_out: Tensor, grad_pose: Optional[Tensor]) -> None:
    Cn = pose.shape[0]
    check(_lib.load().tn_camera_reg(_f32(pose, "pose", (Cn, 6)), Cn, float(trans_pen), float(rot_pen), float(scale), _f32(loss_out, "loss"),
                                    _f32(grad_pose, "grad_pose", (Cn, 6), True), _stream()), "tn_camera_reg")


def train_metrics(losses: Tensor, num_rays: int, thermal_mult: float, poses: Sequence[Tensor], metrics_out: Tensor) -> None:
    """PSNR per spectrum from the pixel-loss sums + the pose norms of up to two camera optimisers, one launch (tn_train_metrics)."""
    if len(poses) > 2 or metrics_out.numel() < 6 or losses.numel() < 6:
        raise ValueError("train_metrics: at most two pose tensors; losses / metrics_out need 6 floats")
    pp = [(_f32(p, "pose", (p.shape[0], 6)), p.shape[0]) for p in poses] + [(None, 0)] * (2 - len(poses))
    check(_lib.load().tn_train_metrics(_f32(losses, "losses"), int(num_rays), float(thermal_mult), pp[0][0], pp[0][1], pp[1][0], pp[1][1],
                                       _f32(metrics_out, "metrics"), _stream()), "tn_train_metrics")


def adam_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999,
              eps: float = 1e-15) -> None:
    n = params.numel()
    for t in (grads, exp_avg, exp_avg_sq):
        if t.numel() != n:
            raise ValueError("Adam arenas must have equal length")
    check(_lib.load().tn_adam_step(_f32(params, "params"), _f32(grads, "grads"), _f32(exp_avg, "exp_avg"), _f32(exp_avg_sq, "exp_avg_sq"), n, int(step),
                                   float(lr), float(beta1), float(beta2), float(eps), _stream()), "tn_adam_step")


def adam_step_ranges(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, ranges, beta1: float = 0.9, beta2: float = 0.999,
                     eps: float = 1e-15) -> None:
    """Adam over several ranges of the same flat arenas in one launch.  ranges: list of (lo, hi, step, lr), element offsets multiples of 4."""
    n = len(ranges)
    if n == 0:
        return
    for t in (grads, exp_avg, exp_avg_sq):
        if t.numel() != params.numel():
            raise ValueError("Adam arenas must have equal length")
    for lo, hi, _, _ in ranges:
        if not (0 <= lo <= hi <= params.numel()):
            raise ValueError(f"Adam range [{lo}, {hi}) outside the arena")
    offs = (C.c_int64 * n)(*[r[0] for r in ranges])
    cnts = (C.c_int64 * n)(*[r[1] - r[0] for r in ranges])
    steps = (C.c_int32 * n)(*[int(r[2]) for r in ranges])
    lrs = (C.c_double * n)(*[float(r[3]) for r in ranges])
    check(_lib.load().tn_adam_step_ranges(_f32(params, "params"), _f32(grads, "grads"), _f32(exp_avg, "exp_avg"), _f32(exp_avg_sq, "exp_avg_sq"), n,
                                          offs, cnts, steps, lrs, float(beta1), float(beta2), float(eps), _stream()), "tn_adam_step_ranges")


def grad_nonfinite(grads: Tensor, found_inf: Tensor) -> None:
    """found_inf[0] = 1 if any element of the (contiguous fp32) gradient slice is inf / NaN; never cleared here."""
    if grads.numel() == 0:
        return
    check(_lib.load().tn_grad_nonfinite(_f32(grads, "grads"), grads.numel(), _f32(found_inf, "found_inf", (1,)), _stream()), "tn_grad_nonfinite")


def grad_nonfinite_ranges(grads: Tensor, ranges, flags, found_inf: Tensor) -> None:
    """One launch: found_inf[flags[k]] = 1 if grads[lo_k:hi_k] holds an inf / NaN (ranges: (lo, hi) element offsets, multiples of 4)."""
    n = len(ranges)
    if n == 0:
        return
    offs = (C.c_int64 * n)(*[int(r[0]) for r in ranges])
    cnts = (C.c_int64 * n)(*[int(r[1] - r[0]) for r in ranges])
    fl = (C.c_int32 * n)(*[int(x) for x in flags])
    check(_lib.load().tn_grad_nonfinite_ranges(_f32(grads, "grads"), n, offs, cnts, fl, int(found_inf.numel()), _f32(found_inf, "found_inf"), _stream()),
          "tn_grad_nonfinite_ranges")


def adam_step_ranges_amp(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, ranges, beta1: float = 0.9, beta2: float = 0.999,
                         eps: float = 1e-15, inv_scale: Optional[Tensor] = None, found_inf: Optional[Tensor] = None, flags=None,
                         skipped: Optional[Tensor] = None, lag_index: int = -1, count_skip: bool = False, schedule=None, sched_step: int = 0,
                         zero_grads: bool = False, scaler_update=None) -> None:
    """adam_step_ranges with GradScaler's skip / unscale decision on the device (no host sync): see tn_adam_step_ranges_amp.
    found_inf: float device tensor with one entry per parameter group; flags: the entry of each range (default 0); skipped: int32 device
    tensor (per-group skip counts, and the schedule lag at lag_index).
    schedule: None, or one (lr_final, max_steps) per range: the range's lr is then lr_init and the exponential-decay schedule is evaluated on the
    device at sched_step - skipped[lag_index].  zero_grads: the launch consumes the gradients (zero behind the read, skipped steps included).
    scaler_update: None, or (scale, growth_tracker, done_counter, growth_factor, backoff_factor, growth_interval): GradScaler.update() by the launch's
    last block (tn_adam_step_ranges_amp_update) instead of a launch of its own."""
    n = len(ranges)
    if n == 0:
        return
    for lo, hi, _, _ in ranges:
        if not (0 <= lo <= hi <= params.numel()):
            raise ValueError(f"Adam range [{lo}, {hi}) outside the arena")
    nflags = int(found_inf.numel()) if found_inf is not None else 1
    if skipped is not None and (skipped.dtype != torch.int32 or not skipped.is_cuda or skipped.numel() < max(nflags, lag_index + 1)):
        raise ValueError("skipped must be an int32 device tensor with an entry per flag (and the lag entry)")
    offs = (C.c_int64 * n)(*[r[0] for r in ranges])
    cnts = (C.c_int64 * n)(*[r[1] - r[0] for r in ranges])
    steps = (C.c_int32 * n)(*[int(r[2]) for r in ranges])
    lrs = (C.c_double * n)(*[float(r[3]) for r in ranges])
    lrf = (C.c_double * n)(*[float(x[0]) for x in schedule]) if schedule is not None else None
    smax = (C.c_int32 * n)(*[int(x[1]) for x in schedule]) if schedule is not None else None
    fl = (C.c_int32 * n)(*[int(x) for x in flags]) if flags is not None else None
    if scaler_update is not None:
        sc, gt, done, gf, bf, gi = scaler_update
        check(_lib.load().tn_adam_step_ranges_amp_update(_f32(params, "params"), _f32(grads, "grads"), _f32(exp_avg, "exp_avg"), _f32(exp_avg_sq, "exp_avg_sq"), n,
                                                         offs, cnts, steps, lrs, lrf, smax, int(sched_step), float(beta1), float(beta2), float(eps),
                                                         _f32(inv_scale, "inv_scale", (1,), True) if inv_scale is not None else None,
                                                         _f32(found_inf, "found_inf"), fl, nflags,
                                                         C.c_void_p(skipped.data_ptr()) if skipped is not None else None, int(lag_index), 1 if count_skip else 0,
                                                         1 if zero_grads else 0, _f32(sc, "scale", (1,)), C.c_void_p(gt.data_ptr()), C.c_void_p(done.data_ptr()),
                                                         float(gf), float(bf), int(gi), _stream()), "tn_adam_step_ranges_amp_update")
        return
    check(_lib.load().tn_adam_step_ranges_amp(_f32(params, "params"), _f32(grads, "grads"), _f32(exp_avg, "exp_avg"), _f32(exp_avg_sq, "exp_avg_sq"), n,
                                              offs, cnts, steps, lrs, lrf, smax, int(sched_step), float(beta1), float(beta2), float(eps),
                                              _f32(inv_scale, "inv_scale", (1,), True) if inv_scale is not None else None,
                                              _f32(found_inf, "found_inf", None, True) if found_inf is not None else None, fl, nflags,
                                              C.c_void_p(skipped.data_ptr()) if skipped is not None else None, int(lag_index), 1 if count_skip else 0,
                                              1 if zero_grads else 0, _stream()), "tn_adam_step_ranges_amp")


def grad_scaler_update(scale: Tensor, growth_tracker: Tensor, found_inf: Tensor, lag: Optional[Tensor], growth_factor: float, backoff_factor: float,
                       growth_interval: int, clear: bool = True) -> None:
    """GradScaler.update() on the device; lag (1-element int32 view or None) += 1 when any found_inf entry is set; clear: zero found_inf after."""
    check(_lib.load().tn_grad_scaler_update(_f32(scale, "scale", (1,)), C.c_void_p(growth_tracker.data_ptr()), _f32(found_inf, "found_inf"),
                                            int(found_inf.numel()), C.c_void_p(lag.data_ptr()) if lag is not None else None, float(growth_factor),
                                            float(backoff_factor), int(growth_interval), 1 if clear else 0, _stream()), "tn_grad_scaler_update")


# ------------------------------------------------------------------------------------------------ 8e: the exchange through the C ABI alone
class RcclComm:
    """An RCCL communicator created through the C ABI (tn_comm_*): what a compiled trainer that binds only include/thermal_nerf_hip.h would use
    for the reference's DistributedDataParallel exchange (pipelines/base_pipeline.py:281-283).  `unique_id` = RcclComm.unique_id() of rank 0,
    handed to every rank by the caller (a file, a socket, torch.distributed's store); creation is collective over the ranks and binds the
    current device.  The package's own training paths exchange through torch.distributed (parallel.py): the same library underneath."""

    def __init__(self, unique_id: bytes, world_size: int, rank: int):
        if len(unique_id) != 128:
            raise ValueError("unique_id is the 128 bytes of RcclComm.unique_id()")
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, 128)
        check(_lib.load().tn_comm_create(buf, int(world_size), int(rank), C.byref(self._h)), "tn_comm_create")
        self.world_size, self.rank = int(world_size), int(rank)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.load().tn_comm_unique_id(buf), "tn_comm_unique_id")
        return buf.raw

    def allreduce_grads(self, grads: Tensor, average: bool = True) -> None:
        """in place over the (contiguous fp32) slice `grads` of the gradient arena, on the current stream"""
        if grads.dtype != torch.float32 or not grads.is_contiguous() or not grads.is_cuda:
            raise ValueError("allreduce_grads takes a contiguous fp32 device tensor (no CPU fallback)")
        check(_lib.load().tn_allreduce_grads(self._h, C.c_void_p(grads.data_ptr()), grads.numel(), 1 if average else 0, _stream()), "tn_allreduce_grads")

    def destroy(self) -> None:
        if self._h:
            check(_lib.load().tn_comm_destroy(self._h), "tn_comm_destroy")
            self._h = C.c_void_p()
