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


# golden sets produced by oracle/make_golden.py from the reference itself: name -> (table-size overrides, rays, file suffix)
SIZES = {"tiny": (TINY, GOLDEN_RAYS, ""), "default": ({}, 64, "_default"), "default256": ({}, 256, "_default256")}


def size_cfg(size, mode="shared", **kw):
    return orc.OracleConfig(density_mode=mode, **{**SIZES[size][0], **kw})


def golden_file(golden_dir, mode, size="tiny"):
    return np.load(os.path.join(golden_dir, f"model_{mode}{SIZES[size][2]}.npz"))


def make_params(cfg, seed=SEED, requires_grad=False):
    p = {k: torch.from_numpy(v) for k, v in synth.synth_params(orc.param_shapes(cfg), seed=seed).items()}
    if requires_grad:
        for v in p.values():
            v.requires_grad_(True)
    return p


def golden_inputs(golden_dir, size="tiny"):
    n = SIZES[size][1]
    if size == "tiny":
        g = np.load(os.path.join(golden_dir, "raygen.npz"))
        rays = {k: g[k] for k in ("origins", "directions", "camera_indices")}
    else:  # the default-size set carries its own rays
        g = golden_file(golden_dir, "shared", size)
        rays = {k: g[f"rays/{k}"] for k in ("origins", "directions", "camera_indices")}
    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, n)
    img, is_th = synth.synth_gt(idx, cams)
    return {
        "cams": cams, "ray_indices": idx, "image": torch.from_numpy(img), "is_thermal": torch.from_numpy(is_th),
        "origins": torch.from_numpy(rays["origins"]), "directions": torch.from_numpy(rays["directions"]),
        "camera_indices": torch.from_numpy(rays["camera_indices"])[:, 0],
        "jitters": [torch.from_numpy(j) for j in synth.synth_jitters(n)],
        "jitters_thermal": [torch.from_numpy(j) for j in synth.synth_jitters(n, tag="_thermal")],
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


_SENS_CACHE = {}


def oracle_grad_sensitivity(golden_dir, mode):
    """Conditioning of the chained train step: how far the ORACLE's own parameter gradients move (relative to each tensor's max |grad|,
    on the golden's sampled indices) when the sampler's jitter input moves by one fp32 ulp.  The two PDF resamplings put the sample
    positions on a piecewise-linear hash grid whose gradient is piecewise CONSTANT in position, so a 1-ulp shift flips cell membership
    for a few samples and the gradients of the main field / pose move by percents (scripts/diag_grad_conditioning.py).  Chained
    gradient comparisons are therefore toleranced by a multiple of this measured sensitivity; the strict bounds live in the op-level
    tests, which feed identical sample positions to both sides."""
    if mode in _SENS_CACHE:
        return _SENS_CACHE[mode]
    g = np.load(os.path.join(golden_dir, f"model_{mode}.npz"))
    cfg = tiny_cfg(mode)
    gi = golden_inputs(golden_dir)

    def run(perturb):
        params = make_params(cfg, requires_grad=True)
        nxt = lambda j: torch.nextafter(j, torch.tensor(2.0))  # noqa: E731
        jit = [nxt(j) for j in gi["jitters"]] if perturb else gi["jitters"]
        jit_t = [nxt(j) for j in gi["jitters_thermal"]] if perturb else gi["jitters_thermal"]
        out = orc.get_outputs(params, cfg, gi["origins"], gi["directions"], gi["camera_indices"], training=True,
                              anneal=float(g["train/anneal"]), jitters=jit, jitters_thermal=jit_t)
        losses = orc.loss_dict(params, cfg, out, gi["image"], gi["is_thermal"], training=True)
        sum(losses.values()).backward()
        return params

    p0, p1 = run(False), run(True)
    sens = {}
    for k in p0:
        if p0[k].grad is None or f"grad_idx/{k}" not in g.files:
            continue
        ii = torch.from_numpy(g[f"grad_idx/{k}"])
        a, b = p0[k].grad.reshape(-1)[ii], p1[k].grad.reshape(-1)[ii]
        sens[k] = (float((a - b).abs().max()) / max(float(a.abs().max()), 1e-12),
                   abs(float(p0[k].grad.norm() - p1[k].grad.norm())) / max(float(p0[k].grad.norm()), 1e-12))
    _SENS_CACHE[mode] = sens
    return sens


def pixel_batch(golden_dir):
    """The jagged image batch of tests/golden/pixels.npz (oracle/make_golden_pixels.py): images in batch order, is_thermal by batch
    position, image_idx = camera of each position, the uniforms the sampler draws, and the reference's outputs."""
    import os

    g = np.load(os.path.join(golden_dir, "pixels.npz"))
    cams = synth.synth_cameras()
    imgs = synth.synth_images(cams)
    order = g["batch_order"].astype(np.int64)
    n = int(g["num_rays"])
    per = (n // len(order)) // 4
    return {
        "images": [torch.from_numpy(imgs[c]) for c in order],
        "is_thermal": torch.from_numpy(cams["is_thermal"][order].astype(np.float32)),
        "image_idx": torch.from_numpy(order),
        "u": torch.from_numpy(synth.synth_patch_uniforms(per * len(order))),
        "num_rays": n,
        "ref": {k: torch.from_numpy(g[k]) for k in ("indices", "image", "is_thermal")},
    }


_DSENS_CACHE = {}


def oracle_density_sensitivity(golden_dir, mode="shared", size="tiny", ulps=1):
    """Conditioning of the chained density: how far the ORACLE's own eval-mode density moves when its level-2 (field) sample bins move by `ulps`
    fp32 ulps, up and down (no GPU involved).  After two PDF resamplings the field samples sit on a piecewise-linear hash grid whose tables are
    deliberately high-variance in the synthetic weights: a 1-ulp shift of a sample changes the interpolated features, the 64-wide MLP amplifies
    it, exp() turns it into density.  Returns per branch suffix {"max": max |d density|, "frac": share of samples that move by more than 1e-4,
    "scale": max density}.  The chained GPU-vs-reference density bound is a stated multiple of this (tests/test_model_gpu.py)."""
    key = (mode, size, ulps)
    if key in _DSENS_CACHE:
        return _DSENS_CACHE[key]
    cfg = size_cfg(size, mode)
    gi = golden_inputs(golden_dir, size)
    params = make_params(cfg)
    with torch.no_grad():
        ref = orc.get_outputs(params, cfg, gi["origins"], gi["directions"], gi["camera_indices"], training=False)
        res = {}
        for sfx, prefix in (("", "field"), ("_thermal", "field_thermal")):
            if f"density{sfx}" not in ref or (sfx and mode != "separate"):
                continue
            smp = ref["samples_list" + sfx][2]
            worst = torch.zeros_like(ref[f"density{sfx}"])
            for toward in (float("inf"), -float("inf")):
                e2p = smp.e_bins
                for _ in range(ulps):
                    e2p = torch.nextafter(e2p, torch.tensor(toward))
                moved = orc.Samples(s_bins=smp.s_bins, e_bins=e2p)
                dp = orc.field_density(params, prefix, cfg, moved.positions(gi["origins"], gi["directions"]))[0]
                worst = torch.maximum(worst, (dp - ref[f"density{sfx}"]).abs())
            res[sfx] = {"max": float(worst.max()), "frac": float((worst > 1e-4).double().mean()), "scale": float(ref[f"density{sfx}"].max())}
    _DSENS_CACHE[key] = res
    return res
