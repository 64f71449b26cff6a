import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from helpers import golden_inputs, make_params, tiny_cfg, sample_indices
from test_model_gpu import build, dev_inputs
gd = os.path.join(ROOT, "tests", "golden")
mode="shared"
g = np.load(os.path.join(gd, f"model_{mode}.npz"))
ocfg, cfg, arena, eng = build(mode)
gi, o, d, cam = dev_inputs(gd)
jit = [j.cuda().reshape(-1).contiguous() for j in gi["jitters"]]
eng.set_anneal_for_step(500)
arena.zero_grad()
out, br = eng.get_outputs(o, d, cam, True, jit, None)
eng.loss_and_backward(out, br, cam, gi["image"].cuda(), gi["is_thermal"].cuda())
name="proposal_networks.0.mlp_base.0.hash_table"
ii = torch.from_numpy(sample_indices(name, arena.view(name).numel()))
assert np.array_equal(ii.numpy(), g[f"grad_idx/{name}"])
p0 = arena.view(name).reshape(-1)[ii.cuda()].cpu().clone()
gr = arena.grad_view(name).reshape(-1)[ii.cuda()].cpu().clone()
eng.optimizer_step(scheduled=False)
p1 = arena.view(name).reshape(-1)[ii.cuda()].cpu()
ref_g = torch.from_numpy(g[f"grad_val/{name}"]); ref_p = torch.from_numpy(g[f"adam_val/{name}"])
diff = (p1 - ref_p).abs()
bad = (diff > 2e-5).nonzero().flatten()
print("n bad", len(bad), "of", len(ii))
for b in bad[:25]:
    print(f"idx {int(ii[b]):8d} g_ref {float(ref_g[b]): .3e} g_hip {float(gr[b]): .3e}  dp_ref {float(ref_p[b]-p0[b]): .3e} dp_hip {float(p1[b]-p0[b]): .3e}")
print("nonzero ref grads:", int((ref_g!=0).sum()), "nonzero hip grads:", int((gr!=0).sum()))
