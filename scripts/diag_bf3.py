"""TN_HEAD_BF16X3 diagnostics: per-parameter gradient error of the split-bf16 head against the oracle and against the fp32 path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import thermal_nerfacto_oracle as orc
from helpers import SEED
from nerfstudio_thermal_amd import ops, synth
from nerfstudio_thermal_amd.netparams import field_params
from test_hip_ops_gpu import DEV, g, md, rays, sample_level, setup_pair

mode = sys.argv[1] if len(sys.argv) > 1 else "shared"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 301
ocfg, params, cfg, arena = setup_pair(mode)
S = 48
r = rays(N)
cam = torch.arange(N) % ocfg.num_images
nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
s, e = sample_level(N, S, nears, fars)
smp = orc.Samples(s_bins=s, e_bins=e)
prefix = "field_thermal" if mode == "separate" else "field"
p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
o = r["origins"].clone().requires_grad_(True)
d = r["directions"].clone().requires_grad_(True)
dens, geo, _, _ = orc.field_density(p, prefix, ocfg, smp.positions(o, d))
rgb = orc.field_color(p, prefix, ocfg, d.detach(), geo, cam, True)
fld = field_params(arena, prefix, cfg, with_grads=True)
C = fld.num_channels
gd = torch.from_numpy(synth.uniform("gbd", (N, S, 1), seed=SEED))
gc = torch.from_numpy(synth.uniform("gbc", (N, S, C), seed=SEED))
((dens * gd).sum() + (rgb * gc).sum()).backward()
k = orc.field_keys(prefix)
keys = ("table", "w0", "b0", "w1", "b1", "hw0", "hb0", "hw1", "hb1", "hw2", "hb2", "emb")
res = {}
for flag in ("0", "1", "01"):
    os.environ["TN_HEAD_BF16X3"] = flag[0]
    arena.zero_grad()
    d_o, d_d = torch.zeros((N, 3), device=DEV), torch.zeros((N, 3), device=DEV)
    ops.field_fwd(fld, g(r["origins"]), g(r["directions"]), g(cam), g(e), True)
    os.environ["TN_HEAD_BF16X3"] = flag[-1]
    ops.field_bwd(fld, g(r["origins"]), g(r["directions"]), g(cam), g(e), g(gd[..., 0]), g(gc), d_o, d_d)
    torch.cuda.synchronize()
    res[flag] = {short: arena.grad_view(k[short]).detach().clone() for short in keys}
    for nm, got, ref in (("d_o", d_o, o.grad), ("d_d", d_d, d.grad)):
        err = (got.cpu() - ref).abs().max(dim=1).values
        print(f"[{flag}] {nm}: max err {float(err.max()):.4g} at ray {int(err.argmax())} of {N}, scale {float(ref.abs().max()):.4g}; rays above 1e-3 of scale: {(err > 1e-3 * float(ref.abs().max())).nonzero().flatten().tolist()[:10]}")
for short in keys:
    ref = p[k[short]].grad
    sc = float(ref.abs().max())
    print(f"{short:6s} scale {sc:10.4g}  fp32 vs oracle {md(res['0'][short], ref)/sc:9.2e}  bf3 vs oracle {md(res['1'][short], ref)/sc:9.2e}  bf3 vs fp32 {md(res['1'][short], res['0'][short])/sc:9.2e}  fwd fp32 + bwd bf3 vs fp32 {md(res['01'][short], res['0'][short])/sc:9.2e}")
