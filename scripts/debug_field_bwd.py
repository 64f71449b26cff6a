"""Debug: where does tn_field_bwd's table gradient differ most from the oracle at production size, and who is closer to float64?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_fullsize_parity_gpu as T
import thermal_nerfacto_oracle as orc
from nerfstudio_thermal_amd import ops, synth
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
from nerfstudio_thermal_amd.netparams import field_params
ocfg = orc.OracleConfig(density_mode="shared")
shapes = {k: v for k, v in orc.param_shapes(ocfg).items() if k.startswith("field.")}
params = {k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, seed=0).items()}
cfg = ThermalNerfactoModelConfig(density_mode="shared")
arena = ParamArena(cfg, ocfg.num_images, "cuda"); arena.load(params)
S = 48
o0, d0, cam = T.patch_rays()
_, e = T.resampled_bins(T.N_RAYS, S, "field")
smp = orc.Samples(s_bins=e, e_bins=e)
gd = torch.from_numpy(synth.uniform("fs_gd", (T.N_RAYS, S, 1), seed=0)) * 1e-2
gc = torch.from_numpy(synth.uniform("fs_gc", (T.N_RAYS, S, 4), seed=0))
k = orc.field_keys("field")
def run(dt):
    p = {kk: v.clone().to(dt).requires_grad_(True) for kk, v in params.items()}
    dens, geo, pre, _ = orc.field_density(p, "field", ocfg, smp.positions(o0.to(dt), d0.to(dt)).to(dt))
    rgb = orc.field_color(p, "field", ocfg, d0.to(dt), geo, cam, True)
    ((dens * gd.to(dt)).sum() + (rgb * gc.to(dt)).sum()).backward()
    return p[k["table"]].grad, dens
g32, dens32 = run(torch.float32)
fld = field_params(arena, "field", cfg, with_grads=True)
hd, hrgb, hpre = ops.field_fwd(fld, T.g(o0), T.g(d0), T.g(cam), T.g(e), True, want_pre=True)
arena.zero_grad()
ops.field_bwd(fld, T.g(o0), T.g(d0), T.g(cam), T.g(e), T.g(gd[..., 0]), T.g(gc), None, None)
got = arena.grad_view(k["table"]).cpu()
dif = (got - g32).abs()
print("hip vs oracle32: max abs", float(dif.max()), "scale", float(g32.abs().max()), "mean abs", float(dif.mean()), "entries > 1e-3:", int((dif > 1e-3).sum()))
w = torch.topk(dif.reshape(-1), 12).indices
# per-slot sum of |contributions| from the oracle: run the backward of sum(|...|)?  cheap proxy: count of touching samples via hash of corner slots
for i in w.tolist():
    r, c = divmod(i, 2)
    print("   level", r >> 19, "slot", r & (2**19 - 1), "f", c, "oracle", float(g32[r, c]), "hip", float(got[r, c]), "diff", float(dif[r, c]))
for lvl in range(16):
    sl = slice(lvl << 19, (lvl + 1) << 19)
    print("level", lvl, "max |diff|", float(dif[sl].max()), "max |grad|", float(g32[sl].abs().max()))
