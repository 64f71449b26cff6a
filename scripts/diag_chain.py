"""Diagnostic: where does the chained-pipeline density error come from? (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import thermal_nerfacto_oracle as orc
from helpers import golden_inputs, make_params, tiny_cfg
from test_hip_ops_gpu import pkg_cfg, md
from nerfstudio_thermal_amd import ops
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.engine import RenderEngine

gd = os.path.join(ROOT, "tests", "golden")
ocfg = tiny_cfg("shared"); cfg = pkg_cfg(ocfg)
params = make_params(ocfg)
arena = ParamArena(cfg, 8, "cuda"); arena.load(params)
eng = RenderEngine(cfg, arena, 8, list(ocfg.is_thermal_cam))
gi = golden_inputs(gd)
o, d, cam = (gi[k].cuda().contiguous() for k in ("origins", "directions", "camera_indices"))
with torch.no_grad():
    ref = orc.get_outputs(params, ocfg, gi["origins"], gi["directions"], gi["camera_indices"], training=False)
out, br = eng.get_outputs(o, d, cam, training=False)
for i, L in enumerate(br[""].levels):
    rs = ref["samples_list"][i]
    print("level", i, "s_bins maxdiff", md(L.s_bins, rs.s_bins), "e_bins max rel", float(((L.e_bins.cpu() - rs.e_bins).abs() / rs.e_bins.clamp_min(1e-6)).max()))
    if i < 2:
        print("   weights maxdiff", md(L.weights, ref["weights_list"][i][..., 0]), "density maxdiff", md(L.density, None) if False else "")
dens_err = (out["density"].cpu() - ref["density"]).abs()
print("chain density err: max", float(dens_err.max()), "frac>1e-4", float((dens_err > 1e-4).float().mean()), "median", float(dens_err.median()))
# field on the oracle's own level-2 bins
e2 = ref["samples_list"][2].e_bins.cuda().contiguous()
hd, hrgb, hpre = ops.field_fwd(eng.field, o, d, cam, e2, False, want_pre=True)
print("same-bins density err", md(hd, ref["density"][..., 0]), "pre err", md(hpre, ref["density_before_activation"][..., 0]), "rgb err", md(hrgb, ref["field_rgb"]))
# sensitivity: oracle density under a 1-ulp perturbation of its own bins
e2p = torch.nextafter(ref["samples_list"][2].e_bins, torch.tensor(float("inf")))
smp = orc.Samples(s_bins=ref["samples_list"][2].s_bins, e_bins=e2p)
with torch.no_grad():
    dp = orc.field_density(params, "field", ocfg, smp.positions(gi["origins"], gi["directions"]))[0]
print("oracle density change under +1ulp bins:", md(dp, ref["density"]))
print("rgb chain err", md(out["rgbt"], torch.cat([ref["rgb"], ref["rgb_thermal"]], -1)))
