"""Shared builders for tests: synthetic params / rays exactly as oracle/make_golden.py used them."""
import os

import numpy as np
import torch

import nerfstudio_thermal_amd  # noqa: F401
from nerfstudio_thermal_amd import synth
import thermal_nerfacto_oracle as orc

TINY = dict(log2_hashmap_size=12, prop_log2_hashmap_size=10)
GOLDEN_RAYS = 32
SEED = 0


def tiny_cfg(mode="shared", **kw):
    return orc.OracleConfig(density_mode=mode, **{**TINY, **kw})


def make_params(cfg, seed=SEED, requires_grad=False):
    p = {k: torch.from_numpy(v) for k, v in synth.synth_params(orc.param_shapes(cfg), seed=seed).items()}
    if requires_grad:
        for v in p.values():
            v.requires_grad_(True)
    return p


def golden_inputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "raygen.npz"))
    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, GOLDEN_RAYS)
    img, is_th = synth.synth_gt(idx, cams)
    return {
        "cams": cams, "ray_indices": idx, "image": torch.from_numpy(img), "is_thermal": torch.from_numpy(is_th),
        "origins": torch.from_numpy(g["origins"]), "directions": torch.from_numpy(g["directions"]),
        "camera_indices": torch.from_numpy(g["camera_indices"])[:, 0],
        "jitters": [torch.from_numpy(j) for j in synth.synth_jitters(GOLDEN_RAYS)],
        "jitters_thermal": [torch.from_numpy(j) for j in synth.synth_jitters(GOLDEN_RAYS, tag="_thermal")],
    }


def maxdiff(a, b):
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).abs().max())


def sample_indices(name, numel, k=2048):
    """Same deterministic index sampler as oracle/make_golden.py."""
    import zlib

    if numel <= k:
        return np.arange(numel, dtype=np.int64)
    return (synth.splitmix64(np.arange(k, dtype=np.uint64) + np.uint64(zlib.crc32(name.encode()))) % np.uint64(numel)).astype(np.int64)
