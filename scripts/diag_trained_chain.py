"""After 200 training iterations on the tiny golden batch: HIP vs oracle on identical samples and through the whole chain, and the ORACLE's own
sensitivity to a 1-ulp move of the rays (how ill-conditioned the chain has become)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import thermal_nerfacto_oracle as orc
from helpers import size_cfg
from test_model_gpu import init_scale_params, dev_inputs, pkg_cfg, DEV
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.engine import RenderEngine
gd = os.path.join(ROOT, "tests", "golden")
for mode in ("shared", "separate"):
    ocfg = size_cfg("tiny", mode); cfg = pkg_cfg(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV); arena.load(init_scale_params(ocfg))
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    gi, o, d, cam = dev_inputs(gd, "tiny")
    img, is_th = gi["image"].to(DEV), gi["is_thermal"].to(DEV)
    for steps in (0, 20, 50, 100, 200):
        while eng.adam_step_count < steps:
            eng.train_step(o, d, cam, img, is_th, eng.adam_step_count)
        params = {k: arena.view(k).detach().cpu().clone() for k in arena.names()}
        with torch.no_grad():
            ref = orc.get_outputs(params, ocfg, gi["origins"], gi["directions"], gi["camera_indices"], training=False, anneal=eng.anneal)
            ref2 = orc.get_outputs(params, ocfg, torch.nextafter(gi["origins"], torch.tensor(9.0)), gi["directions"], gi["camera_indices"], training=False, anneal=eng.anneal)
        out, br = eng.get_outputs(o, d, cam, training=False)
        b = br[""]
        e2 = b.levels[2].e_bins.cpu()
        with torch.no_grad():
            pos = orc.Samples(s_bins=e2, e_bins=e2).positions(b.origins.cpu(), b.directions.cpu())
            dref = orc.field_density(params, "field", ocfg, pos)[0][..., 0]
        err = (b.levels[2].density.cpu() - dref).abs() / dref.abs().clamp(min=1.0)
        print(mode, steps, "dens max", float(dref.max()), "| identical-sample density rel err", float(err.max()),
              "| chain rgb err HIP-vs-oracle", float((out["rgb"].cpu() - ref["rgb"]).abs().max()),
              "| oracle-vs-oracle(1 ulp) rgb", float((ref2["rgb"] - ref["rgb"]).abs().max()),
              "| chain dens err", float((out["density"].cpu() - ref["density"]).abs().max()), "oracle 1ulp dens", float((ref2["density"] - ref["density"]).abs().max()))
