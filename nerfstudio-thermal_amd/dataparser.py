"""RGB+T `transforms.json` datasets on disk: reader, writer and loader.

Reader  = ThermalNerf dataparser (data/dataparsers/thermalnerf_dataparser.py:17-30 over nerfstudio_dataparser.py:67-466): frames sorted by
          file name, per-frame or global intrinsics and distortion, `is_thermal` per frame, "up" orientation + centring + auto scale of the
          poses (cameras/camera_utils.py:449-474,515-623), the thermal-aware train/eval split (data/utils/dataparsers_utils.py:23-73).
Writer  = the layout process_data/rgbt_to_nerfstudio_dataset.py:240-266 produces: RGB frames first, then the thermal frames, every frame with
          its own fl_x/fl_y/cx/cy/w/h/k1/k2/p1/p2 and `is_thermal`, images under images/ and images_thermal/ (8-bit PNG; thermal single channel).
Loader  = InputDataset.get_image_float32 (data/datasets/base_dataset.py:61-95): uint8 / 255, a single-channel image repeated to 3 channels.

The outputs feed ops.ImageCache / DeviceDataManager (the per-step pixel sampling and ray generation run on the device) and
ThermalNerfactoModel.get_outputs_for_camera.  Host-side, one-off work: plain numpy / torch CPU; nothing here is on the per-ray path.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor


@dataclass
class ThermalNerfDataParserConfig:
    data: str = ""
    scale_factor: float = 1.0
    downscale_factor: Optional[int] = None
    scene_scale: float = 1.0
    orientation_method: str = "up"  # "up" | "none"
    center_method: str = "poses"  # "poses" | "none"
    auto_scale_poses: bool = True
    eval_mode: str = "fraction"  # "fraction" | "interval" | "all"
    train_split_fraction: float = 0.9
    eval_interval: int = 8

    def setup(self) -> "ThermalNerf":
        return ThermalNerf(self)


@dataclass
class DataparserOutputs:
    image_filenames: List[str]
    cameras: Dict[str, Tensor]  # c2w [C,3,4], fx, fy, cx, cy [C] fp32, width, height [C] int32, distortion [C,6] (k1,k2,k3,k4,p1,p2)
    scene_box_aabb: Tensor  # [2,3]
    dataparser_scale: float
    dataparser_transform: Tensor  # [3,4]
    metadata: Dict[str, list] = field(default_factory=dict)  # "is_thermal": per image


def rotation_matrix(a: Tensor, b: Tensor) -> Tensor:
    """Rotation taking direction a to direction b (cameras/camera_utils.py:449-474; the exactly-opposite case is nudged deterministically)."""
    a = a / torch.linalg.norm(a)
    b = b / torch.linalg.norm(b)
    v = torch.linalg.cross(a, b)
    c = torch.dot(a, b)
    if c < -1 + 1e-8:
        return rotation_matrix(a + torch.tensor([1e-3, -2e-3, 1.5e-3]), b)
    s = torch.linalg.norm(v)
    k = torch.tensor([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    return torch.eye(3) + k + k @ k * ((1 - c) / (s**2 + 1e-8))


def auto_orient_and_center_poses(poses: Tensor, method: str = "up", center_method: str = "poses") -> Tuple[Tensor, Tensor]:
    """cameras/camera_utils.py:515-623 for the methods the thermal pipeline uses.  poses [C,4,4] -> ([C,3,4], transform [3,4])."""
    origins = poses[..., :3, 3]
    mean_origin = torch.mean(origins, dim=0)
    if center_method == "poses":
        translation = mean_origin
    elif center_method == "none":
        translation = torch.zeros_like(mean_origin)
    else:
        raise NotImplementedError(f"center_method={center_method}")
    if method == "up":
        up = torch.mean(poses[:, :3, 1], dim=0)
        up = up / torch.linalg.norm(up)
        rot = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rot, rot @ -translation[..., None]], dim=-1)
    elif method == "none":
        transform = torch.eye(4)
        transform[:3, 3] = -translation
        transform = transform[:3, :]
    else:
        raise NotImplementedError(f"orientation_method={method}")
    return transform @ poses, transform


def train_eval_split_fraction(image_filenames: Sequence[str], train_split_fraction: float) -> Tuple[np.ndarray, np.ndarray]:
    """data/utils/dataparsers_utils.py:23-73: equally spaced training images; for a thermal dataset (file names under images_thermal/) the
    same frame numbers are chosen in both spectra (the list holds the RGB block first, then the thermal block, as the sort by name leaves it)."""
    total = len(image_filenames)
    num_thermal = sum("images_thermal" in str(f) for f in image_filenames)
    num_rgb = total - num_thermal
    n = min(num_rgb, num_thermal) if num_thermal > 0 else total
    n_train = math.ceil(n * train_split_fraction)
    i_train = np.linspace(0, n - 1, n_train, dtype=int)
    i_eval = np.setdiff1d(np.arange(n), i_train)
    if num_thermal > 0:
        rem = max(num_rgb, num_thermal) - n
        r_train = np.linspace(0, rem - 1, math.ceil(rem * train_split_fraction), dtype=int)
        r_eval = np.setdiff1d(np.arange(rem), r_train)
        r_train, r_eval = r_train + n, r_eval + n
        if n == num_rgb:
            i_train = np.concatenate((i_train, i_train + num_rgb, r_train + num_rgb))
            i_eval = np.concatenate((i_eval, i_eval + num_rgb, r_eval + num_rgb))
        else:
            i_train = np.concatenate((i_train, r_train, i_train + num_rgb))
            i_eval = np.concatenate((i_eval, r_eval, i_eval + num_rgb))
    assert total == len(i_train) + len(i_eval) and len(np.intersect1d(i_train, i_eval)) == 0
    return i_train, i_eval


def _distortion(src: dict) -> List[float]:
    if "distortion_params" in src:
        return [float(x) for x in src["distortion_params"]]
    return [float(src.get(k, 0.0)) for k in ("k1", "k2", "k3", "k4", "p1", "p2")]


class ThermalNerf:
    def __init__(self, config: ThermalNerfDataParserConfig):
        self.config = config
        self.downscale_factor = config.downscale_factor

    def _fname(self, file_path: str, data_dir: str) -> str:
        """_get_fname of both parsers: <folder>_<factor>/<name> when a downscale factor > 1 is in use (thermal frames use their own folder
        name as the prefix), auto factor = smallest power of two that brings the longer side under 1600 px and exists on disk."""
        folder, name = os.path.split(file_path)
        if self.downscale_factor is None:
            from PIL import Image

            w, h = Image.open(os.path.join(data_dir, file_path)).size
            df = 0
            while max(h, w) / 2**df > 1600 and os.path.exists(os.path.join(data_dir, f"{os.path.basename(folder)}_{2 ** (df + 1)}", name)):
                df += 1
            self.downscale_factor = 2**df
        if self.downscale_factor > 1:
            return os.path.join(data_dir, f"{os.path.basename(folder)}_{self.downscale_factor}", name)
        return os.path.join(data_dir, file_path)

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        c = self.config
        path = c.data
        meta_path = path if path.endswith(".json") else os.path.join(path, "transforms.json")
        data_dir = os.path.dirname(meta_path)
        with open(meta_path, encoding="utf-8") as f:
            meta = json.load(f)
        fnames = [self._fname(fr["file_path"], data_dir) for fr in meta["frames"]]
        order = np.argsort(fnames)
        frames = [meta["frames"][i] for i in order]
        image_filenames = [fnames[i] for i in order]

        def per_frame(key, cast):
            return [cast(meta[key]) if key in meta else cast(fr[key]) for fr in frames]

        fx, fy = per_frame("fl_x", float), per_frame("fl_y", float)
        cx, cy = per_frame("cx", float), per_frame("cy", float)
        height, width = per_frame("h", int), per_frame("w", int)
        global_dist = any(k in meta for k in ("k1", "k2", "k3", "p1", "p2", "distortion_params"))
        dist = [_distortion(meta) if global_dist else _distortion(fr) for fr in frames]
        poses = torch.from_numpy(np.array([fr["transform_matrix"] for fr in frames]).astype(np.float32))
        if c.eval_mode == "fraction":
            i_train, i_eval = train_eval_split_fraction(image_filenames, c.train_split_fraction)
        elif c.eval_mode == "interval":
            allv = np.arange(len(image_filenames))
            i_train, i_eval = allv[allv % c.eval_interval != 0], allv[allv % c.eval_interval == 0]
        elif c.eval_mode == "all":
            i_train = i_eval = np.arange(len(image_filenames))
        else:
            raise NotImplementedError(f"eval_mode={c.eval_mode}")
        if split == "train":
            indices = i_train
        elif split in ("val", "test"):
            indices = i_eval
        else:
            raise ValueError(f"Unknown dataparser split {split}")
        poses, transform = auto_orient_and_center_poses(poses, method=meta.get("orientation_override", c.orientation_method), center_method=c.center_method)
        scale = 1.0
        if c.auto_scale_poses:
            scale /= float(torch.max(torch.abs(poses[:, :3, 3])))
        scale *= c.scale_factor
        poses[:, :3, 3] *= scale
        idx = torch.as_tensor(np.asarray(indices), dtype=torch.long)
        s = 1.0 / (self.downscale_factor or 1)  # Cameras.rescale_output_resolution (cameras/cameras.py:930-967)
        t = lambda v, dt=torch.float32: torch.tensor(v, dtype=dt)[idx]  # noqa: E731
        cams = {"c2w": poses[idx][:, :3, :4].contiguous(), "fx": t(fx) * s, "fy": t(fy) * s, "cx": t(cx) * s, "cy": t(cy) * s,
                "width": (t(width) * s).to(torch.int32), "height": (t(height) * s).to(torch.int32), "distortion": t(dist)}
        a = c.scene_scale
        dataparser_transform = transform
        if "applied_transform" in meta:
            at = torch.tensor(meta["applied_transform"], dtype=transform.dtype)
            dataparser_transform = transform @ torch.cat([at, torch.tensor([[0, 0, 0, 1]], dtype=transform.dtype)], 0)
        if "applied_scale" in meta:
            scale *= float(meta["applied_scale"])
        return DataparserOutputs(image_filenames=[image_filenames[i] for i in indices], cameras=cams,
                                 scene_box_aabb=torch.tensor([[-a, -a, -a], [a, a, a]], dtype=torch.float32), dataparser_scale=scale,
                                 dataparser_transform=dataparser_transform, metadata={"is_thermal": [frames[i]["is_thermal"] for i in indices]})


def load_image_float32(path: str) -> Tensor:
    """[H,W,3] fp32 in [0,1] (base_dataset.py:61-95: uint8 / 255; a single-channel image is repeated to three channels; alpha dropped)."""
    from PIL import Image

    im = np.array(Image.open(path), dtype="uint8")
    if im.ndim == 2:
        im = im[:, :, None].repeat(3, axis=2)
    return torch.from_numpy(im[:, :, :3].astype("float32") / 255.0)


def write_rgbt_dataset(out_dir: str, cams: Dict[str, np.ndarray], images: Sequence[np.ndarray]) -> str:
    """Write cameras + images as an RGB+T nerfstudio dataset (the layout of process_data/rgbt_to_nerfstudio_dataset.py:240-266).
    cams: c2w [C,3,4], fx, fy, cx, cy, width, height, distortion [C,6] (k1,k2,k3,k4,p1,p2), is_thermal [C]; images [H,W,3] fp32 in [0,1].
    RGB frames are written first, then the thermal frames; frame k of each spectrum is frame_{k+1:05d}.png."""
    from PIL import Image

    os.makedirs(os.path.join(out_dir, "images"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "images_thermal"), exist_ok=True)
    frames = []
    counters = {0: 0, 1: 0}
    order = [i for i in range(len(images)) if not cams["is_thermal"][i]] + [i for i in range(len(images)) if cams["is_thermal"][i]]
    for i in order:
        th = int(bool(cams["is_thermal"][i]))
        counters[th] += 1
        rel = os.path.join("images_thermal" if th else "images", f"frame_{counters[th]:05d}.png")
        u8 = np.clip(np.rint(np.asarray(images[i]) * 255.0), 0, 255).astype(np.uint8)
        Image.fromarray(u8[:, :, 0] if th else u8).save(os.path.join(out_dir, rel))
        m = np.eye(4)
        m[:3, :4] = cams["c2w"][i]
        d = cams["distortion"][i]
        frames.append({"file_path": rel.replace(os.sep, "/"), "transform_matrix": m.tolist(), "colmap_im_id": len(frames) + 1, "is_thermal": th,
                       "fl_x": float(cams["fx"][i]), "fl_y": float(cams["fy"][i]), "cx": float(cams["cx"][i]), "cy": float(cams["cy"][i]),
                       "w": int(cams["width"][i]), "h": int(cams["height"][i]), "k1": float(d[0]), "k2": float(d[1]), "p1": float(d[4]), "p2": float(d[5])})
    path = os.path.join(out_dir, "transforms.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"camera_model": "OPENCV", "frames": frames}, f, indent=4)
    return path
