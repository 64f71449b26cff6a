"""Deterministic synthetic inputs for parity tests, smoke() and bench.py.

Everything is generated from an integer hash (splitmix64 of the flat element
index, salted by a per-tensor name hash), so the same arrays can be rebuilt on
any machine without shipping them: the committed golden fixtures store only the
*outputs* of the reference on these inputs.

High-variance on purpose: at the reference's own initialisation
(hash_init_scale=1e-3) the field is almost constant and a wrong hash index or a
swapped corner would pass a 1e-3 check (hard part 1).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _salt(name: str, seed: int) -> np.uint64:
    return np.uint64((zlib.crc32(name.encode()) << 20) ^ (seed * 0x1000193))


def uniform(name: str, shape: Tuple[int, ...], lo: float = -1.0, hi: float = 1.0, seed: int = 0) -> np.ndarray:
    """U[lo,hi) float32 array; element i depends only on (name, seed, i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + _salt(name, seed))
    u = (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))  # 24 random bits -> exact in fp32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def synth_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, table_scale: float = 0.5,
                 pose_scale: float = 1e-3) -> Dict[str, np.ndarray]:
    """Weights for every tensor in `shapes` (reference state_dict names).

    hash tables x table_scale, Linear weights x sqrt(6/fan_in), biases x 0.1, appearance embedding x 1,
    pose_adjustment x pose_scale.
    """
    out = {}
    for name, shape in shapes.items():
        if name.endswith("hash_table"):
            out[name] = uniform(name, shape, seed=seed) * np.float32(table_scale)
        elif name.endswith("pose_adjustment"):
            out[name] = uniform(name, shape, seed=seed) * np.float32(pose_scale)
        elif name.endswith("embedding.weight"):
            out[name] = uniform(name, shape, seed=seed)
        elif name.endswith(".weight"):
            out[name] = uniform(name, shape, seed=seed) * np.float32(np.sqrt(6.0 / shape[-1]))
        elif name.endswith(".bias"):
            out[name] = uniform(name, shape, seed=seed) * np.float32(0.1)
        else:
            raise KeyError(name)
    return out


# ------------------------------------------------------------------------------------------
# 8-camera RGB+T scene (4 RGB 640x480 fx=600; 4 thermal 160x120 fx=150) on a circle, looking at the origin
# ------------------------------------------------------------------------------------------


def synth_cameras(num_rgb: int = 4, num_thermal: int = 4, radius: float = 0.8, height: float = 0.3) -> Dict[str, np.ndarray]:
    C = num_rgb + num_thermal
    c2w = np.zeros((C, 3, 4), dtype=np.float32)
    for i in range(C):
        ang = 2.0 * np.pi * i / C
        pos = np.array([radius * np.cos(ang), radius * np.sin(ang), height], dtype=np.float64)
        fwd = -pos / np.linalg.norm(pos)  # camera looks along -z (OpenGL)
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        true_up = np.cross(right, fwd)
        c2w[i, :, 0] = right
        c2w[i, :, 1] = true_up
        c2w[i, :, 2] = -fwd
        c2w[i, :, 3] = pos
    is_thermal = np.array([0] * num_rgb + [1] * num_thermal, dtype=np.int64)
    width = np.where(is_thermal == 1, 160, 640).astype(np.int64)
    heightpx = np.where(is_thermal == 1, 120, 480).astype(np.int64)
    f = np.where(is_thermal == 1, 150.0, 600.0).astype(np.float32)
    dist = np.zeros((C, 6), dtype=np.float32)  # k1 k2 k3 k4 p1 p2
    dist[:, 0] = np.where(is_thermal == 1, -0.08, 0.05)
    dist[:, 1] = np.where(is_thermal == 1, 0.02, -0.01)
    dist[:, 4] = 1e-3
    dist[:, 5] = -5e-4
    return {
        "c2w": c2w, "fx": f, "fy": f.copy(), "cx": (width / 2).astype(np.float32), "cy": (heightpx / 2).astype(np.float32),
        "width": width, "height": heightpx, "distortion": dist, "is_thermal": is_thermal,
    }


def synth_ray_indices(cams: Dict[str, np.ndarray], num_rays: int, seed: int = 42) -> np.ndarray:
    """[N,3] int64 (camera,row,col): N/C rays per camera in camera order, as 2x2 patches whose 4 pixels are adjacent
    (what PatchPixelSampler(patch_size=2) + collate_image_dataset_batch_list produce, data/pixel_samplers.py:296-312,421-438)."""
    C = cams["c2w"].shape[0]
    per_cam = num_rays // C
    assert per_cam % 4 == 0 and per_cam * C == num_rays, "num_rays must be a multiple of 4*num_cameras"
    out = np.zeros((num_rays, 3), dtype=np.int64)
    k = 0
    for c in range(C):
        npatch = per_cam // 4
        H, W = int(cams["height"][c]), int(cams["width"][c])
        r = (uniform(f"patch_row_{c}", (npatch,), 0.0, 1.0, seed) * (H - 1)).astype(np.int64).clip(0, H - 2)
        q = (uniform(f"patch_col_{c}", (npatch,), 0.0, 1.0, seed) * (W - 1)).astype(np.int64).clip(0, W - 2)
        for dy in (0, 1):
            for dx in (0, 1):
                sl = slice(k + dy * 2 + dx, k + 4 * npatch, 4)
                out[sl, 0] = c
                out[sl, 1] = r + dy
                out[sl, 2] = q + dx
        k += 4 * npatch
    return out


def synth_gt(ray_indices: np.ndarray, cams: Dict[str, np.ndarray], seed: int = 7) -> Tuple[np.ndarray, np.ndarray]:
    """Ground-truth pixels [N,3] (thermal stored as grey x3) and is_thermal [N] float32."""
    N = ray_indices.shape[0]
    is_th = cams["is_thermal"][ray_indices[:, 0]].astype(np.float32)
    rgb = uniform("gt_rgb", (N, 3), 0.0, 1.0, seed)
    grey = uniform("gt_thermal", (N, 1), 0.0, 1.0, seed)
    img = np.where(is_th[:, None] > 0, np.repeat(grey, 3, axis=1), rgb).astype(np.float32)
    return img, is_th


def synth_gt_smooth(ray_indices: np.ndarray, cams: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """A learnable target: smooth colour / temperature as a function of the pixel position (per camera phase)."""
    c = ray_indices[:, 0].astype(np.float64)
    y = ray_indices[:, 1] / cams["height"][ray_indices[:, 0]].astype(np.float64)
    x = ray_indices[:, 2] / cams["width"][ray_indices[:, 0]].astype(np.float64)
    is_th = cams["is_thermal"][ray_indices[:, 0]].astype(np.float32)
    rgb = np.stack([0.5 + 0.4 * np.sin(6.0 * x + c), 0.5 + 0.4 * np.cos(5.0 * y - c), 0.5 + 0.3 * np.sin(4.0 * (x + y))], axis=1)
    grey = (0.5 + 0.4 * np.cos(3.0 * x - 2.0 * y + 0.5 * c))[:, None]
    img = np.where(is_th[:, None] > 0, np.repeat(grey, 3, axis=1), rgb).astype(np.float32)
    return img, is_th


def synth_images(cams: Dict[str, np.ndarray]) -> list:
    """Per-camera training images [H,W,3] float32 (thermal frames grey x3, as ThermalDataset stores them): synth_gt_smooth on every pixel."""
    out = []
    for c in range(cams["c2w"].shape[0]):
        H, W = int(cams["height"][c]), int(cams["width"][c])
        yy, xx = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
        idx = np.stack([np.full(H * W, c, dtype=np.int64), yy.reshape(-1), xx.reshape(-1)], axis=1)
        img, _ = synth_gt_smooth(idx, cams)
        out.append(img.reshape(H, W, 3))
    return out


def synth_patch_uniforms(num_patches: int, seed: int = 9, tag: str = "") -> np.ndarray:
    """[num_patches, 3] uniforms in [0,1): what PatchPixelSampler draws per image batch (column 0 is drawn and unused for a single image)."""
    return uniform(f"patch_u{tag}", (num_patches, 3), 0.0, 1.0, seed)


def synth_jitters(num_rays: int, seed: int = 3, tag: str = "") -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The three per-ray uniforms the training sampler draws (level-0 stratified, two PDF levels)."""
    return tuple(uniform(f"jitter{tag}_{i}", (num_rays, 1), 0.0, 1.0, seed) for i in range(3))


def synth_rays_simple(num_rays: int, seed: int = 11) -> Dict[str, np.ndarray]:
    """Rays without a camera model (unit tests of the sampler/field kernels): origins on a 0.8 shell, aimed near the origin."""
    o = uniform("ray_o", (num_rays, 3), -1.0, 1.0, seed)
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * np.float32(0.8)
    tgt = uniform("ray_t", (num_rays, 3), -0.3, 0.3, seed)
    d = tgt - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    cam = (splitmix64(np.arange(num_rays, dtype=np.uint64) + np.uint64(seed)) % np.uint64(8)).astype(np.int64)
    return {"origins": o.astype(np.float32), "directions": d.astype(np.float32), "camera_indices": np.sort(cam)}


def cube_scene_images(origins: np.ndarray, directions: np.ndarray, is_thermal: bool, half: float = 0.35) -> np.ndarray:
    """Analytic RGB / thermal image of a textured cube [-half, half]^3 in front of a direction-dependent background, for rays [P,3]: a
    multi-view CONSISTENT scene ('analytic unit cube with per-face colour / temperature'), unlike synth_images.  -> [P,3] fp32
    (thermal: the temperature repeated three times, as ThermalDataset stores it)."""
    o, d = origins.astype(np.float64), directions.astype(np.float64)
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    t0, t1 = (-half - o) * inv, (half - o) * inv
    tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
    t_in, t_out = tn.max(axis=1), tf.min(axis=1)
    hit = (t_out >= np.maximum(t_in, 0.0)) & (t_in > 0.0)
    axis = tn.argmax(axis=1)
    p = o + d * t_in[:, None]
    sign = (np.take_along_axis(p, axis[:, None], axis=1)[:, 0] > 0).astype(np.int64)
    face = 2 * axis + sign
    uv = np.stack([np.take_along_axis(p, ((axis + 1) % 3)[:, None], 1)[:, 0], np.take_along_axis(p, ((axis + 2) % 3)[:, None], 1)[:, 0]], 1) / half
    tex = 0.8 + 0.2 * np.sin(5.0 * uv[:, 0]) * np.sin(5.0 * uv[:, 1])
    face_rgb = np.array([[0.9, 0.2, 0.2], [0.2, 0.8, 0.3], [0.2, 0.3, 0.9], [0.9, 0.8, 0.2], [0.8, 0.3, 0.8], [0.2, 0.8, 0.8]])
    face_temp = np.array([0.9, 0.3, 0.7, 0.5, 0.8, 0.4])
    if is_thermal:
        bg = 0.25 + 0.1 * d[:, 2]
        val = np.where(hit, face_temp[face] * tex, bg)
        return np.repeat(val[:, None], 3, axis=1).astype(np.float32)
    bg = 0.5 + 0.35 * d
    return np.where(hit[:, None], face_rgb[face] * tex[:, None], bg).astype(np.float32)


# ---- thermal-splatfacto (N4): synthetic Gaussians and a look-at camera (inputs of bench.py --workload splat and of the splat tests)
def synth_gaussians(num: int, seed: int = 0, extent: float = 1.0, scale_range=(-4.5, -2.5)) -> "Dict[str, torch.Tensor]":
    """Deterministic synthetic scene: Gaussians in a cube of half-size `extent`, log-scales uniform in scale_range, random rotations,
    mixed opacities, random SH coefficients (degree-3 RGB + thermal)."""
    g = np.random.default_rng(seed)
    f = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32))  # noqa: E731
    u = lambda lo, hi, *s: torch.from_numpy(g.uniform(lo, hi, s).astype(np.float32))  # noqa: E731
    return {"means": u(-extent, extent, num, 3), "scales": u(scale_range[0], scale_range[1], num, 3), "quats": f(num, 4), "opacities": u(-2.0, 4.0, num, 1),
            "features_dc": f(num, 3) * 0.8, "features_rest": f(num, 15, 3) * 0.15, "features_dc_thermal": f(num, 1) * 0.8,
            "features_rest_thermal": f(num, 15, 1) * 0.15}


def look_at_camera(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)) -> "torch.Tensor":
    """camera-to-world [3,4] in nerfstudio's convention (x right, y up, z back)."""
    eye, target, up = (torch.tensor(v, dtype=torch.float32) for v in (eye, target, up))
    back = eye - target
    back = back / back.norm()
    right = torch.linalg.cross(up, back)
    right = right / right.norm()
    upv = torch.linalg.cross(back, right)
    return torch.stack([right, upv, back, eye], 1)
