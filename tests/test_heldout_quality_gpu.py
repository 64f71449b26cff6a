"""Held-out quality against the oracle's (the 120-iteration training curve pins the TRAIN loss only).

tests/golden/heldout_shared.json (oracle/make_heldout.py): the oracle trains 300 iterations at 1024 rays on the train split of the synthetic RGB+T
cube scene ON DISK -- transforms.json through the dataparser, PatchPixelSampler batches over the jagged image list, per-iteration jitter -- and
renders the HELD-OUT cameras in eval mode (mean appearance embedding, no pose correction: pipelines/base_pipeline.py:377-440,
models/thermal_nerfacto.py:403-489).  The HIP path replays the same schedule: same scene files, same pixel-sampler uniforms (tn_sample_rays is
bit-exact against the reference's sampler), same jitter; its held-out PSNR per spectrum must land where the oracle's does."""
import json
import os

import numpy as np
import pytest
import torch

import heldout_common as hc
import thermal_nerfacto_oracle as orc
from nerfstudio_thermal_amd import ops, synth
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.engine import RenderEngine
from test_hip_ops_gpu import pkg_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_heldout_psnr_matches_the_oracles(golden_dir, tmp_path):
    with open(os.path.join(golden_dir, "heldout_shared.json")) as f:
        gold = json.load(f)
    assert (gold["steps"], gold["num_rays"], gold["frames_per_spectrum"]) == (hc.T_STEPS, hc.N_RAYS, hc.FRAMES)
    hc.write_scene(str(tmp_path))
    tr, tr_images, va, va_images = hc.splits(str(tmp_path))
    is_th = [int(x) for x in tr.metadata["is_thermal"]]
    ocfg = orc.OracleConfig(density_mode="shared", num_images=len(tr_images), is_thermal_cam=tuple(is_th), **hc.TINY)
    cfg = pkg_cfg(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load({k: torch.from_numpy(v) for k, v in synth.synth_params(orc.param_shapes(ocfg), seed=0, table_scale=0.1).items()})
    eng = RenderEngine(cfg, arena, ocfg.num_images, is_th)
    cache = ops.ImageCache.build(tr_images, torch.tensor([float(x) for x in is_th]), torch.arange(len(tr_images)), DEV)
    cams = {k: tr.cameras[k].to(DEV).contiguous() for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    first = last = None
    for step in range(hc.T_STEPS):
        u = torch.from_numpy(hc.step_uniforms(step)).to(DEV).contiguous()
        o, d, cam, img, is_thermal, _ = ops.sample_rays(cache, hc.N_RAYS, u, cams, 2)
        jit = [torch.from_numpy(j).to(DEV).reshape(-1) for j in hc.step_jitters(step)]
        losses = eng.train_step(o, d, cam, img, is_thermal, step, jitters=jit)
        if step in (0, hc.T_STEPS - 1):
            tot = float(sum(losses.values()))
            first, last = (tot, last) if step == 0 else (first, tot)
    assert abs(first - gold["total_loss_first"]) <= 2e-3 * gold["total_loss_first"]  # the same first batch, the same loss
    assert 0.5 <= last / gold["total_loss_last"] <= 2.0, (last, gold["total_loss_last"])
    vcams = {k: va.cameras[k].to(DEV).contiguous() for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    got = {"rgb": [], "thermal": []}
    for i in range(len(va_images)):
        idx = hc.eval_indices(va, i).to(DEV)
        o, d, _, _ = ops.raygen(idx, vcams["c2w"], vcams["fx"], vcams["fy"], vcams["cx"], vcams["cy"], vcams["distortion"])
        out, _ = eng.get_outputs(o, d, torch.zeros(idx.shape[0], dtype=torch.int64, device=DEV), training=False)
        gt = va_images[i][idx[:, 1].cpu(), idx[:, 2].cpu()]
        if va.metadata["is_thermal"][i]:
            got["thermal"].append(hc.psnr(out["rgb_thermal"].cpu(), gt[:, :1]))
        else:
            got["rgb"].append(hc.psnr(out["rgb"].cpu(), gt))
    for key in ("rgb", "thermal"):
        mine, ref, per = float(np.mean(got[key])), gold["heldout_psnr"][key], gold["heldout_psnr_perturbed"][key]
        # within 1.5 dB of the oracle's held-out PSNR (plus the oracle's own spread under one ulp of jitter: training is a chaotic map)
        assert abs(mine - ref) <= 1.5 + abs(per - ref), (key, mine, ref, per)
    print("held-out PSNR (HIP / oracle / oracle, jitter + 1 ulp):", {k: (float(np.mean(got[k])), gold["heldout_psnr"][k], gold["heldout_psnr_perturbed"][k]) for k in got})
