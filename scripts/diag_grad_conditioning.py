"""How much do the REFERENCE-equivalent (oracle) gradients move when the sampler input moves by one fp32 ulp?  (CPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import thermal_nerfacto_oracle as orc
from helpers import golden_inputs, make_params, tiny_cfg

gd = os.path.join(ROOT, "tests", "golden")
mode = sys.argv[1] if len(sys.argv) > 1 else "shared"
g = np.load(os.path.join(gd, f"model_{mode}.npz"))
cfg = tiny_cfg(mode)
gi = golden_inputs(gd)

def run(perturb):
    params = make_params(cfg, requires_grad=True)
    jit = [torch.nextafter(j, torch.tensor(2.0)) if perturb else j for j in gi["jitters"]]
    out = orc.get_outputs(params, cfg, gi["origins"], gi["directions"], gi["camera_indices"], training=True, anneal=float(g["train/anneal"]),
                          jitters=jit, jitters_thermal=gi["jitters_thermal"])
    losses = orc.loss_dict(params, cfg, out, gi["image"], gi["is_thermal"], training=True)
    sum(losses.values()).backward()
    return params, out

p0, o0 = run(False)
p1, o1 = run(True)
print("density change", float((o0["density"] - o1["density"]).abs().max()))
for k in p0:
    if p0[k].grad is None or f"grad_idx/{k}" not in g.files:
        continue
    ii = torch.from_numpy(g[f"grad_idx/{k}"])
    a, b = p0[k].grad.reshape(-1)[ii], p1[k].grad.reshape(-1)[ii]
    scale = float(a.abs().max())
    print(f"{k:60s} sampled rel change {float((a-b).abs().max())/max(scale,1e-12):.3e}  norm rel change {abs(float(p0[k].grad.norm()-p1[k].grad.norm()))/float(p0[k].grad.norm()):.3e}")
