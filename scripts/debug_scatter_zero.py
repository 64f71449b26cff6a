"""Debug: where does the zero pattern of the full-size main-grid scatter differ from the oracle's?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_fullsize_parity_gpu as T
import thermal_nerfacto_oracle as orc
from nerfstudio_thermal_amd import ops, synth
L, log2T, S = 16, 19, 48
o0, d0, _ = T.patch_rays()
_, e = T.resampled_bins(T.N_RAYS, S, "main")
smp = orc.Samples(s_bins=e, e_bins=e)
table = (torch.from_numpy(synth.uniform("fs_table", (L * 2**log2T, 2), seed=0)) * 0.5).requires_grad_(True)
res = orc.level_resolutions(L, 16, 2048)
p, _ = orc.unit_cube_positions(smp.positions(o0, d0))
enc = orc.hash_encode(p.view(-1, 3), table, res, log2T)
g_enc = torch.from_numpy(synth.uniform("fs_g", (T.N_RAYS * S, 2 * L), seed=0))
(enc * g_enc).sum().backward()
tg = torch.zeros((L * 2**log2T, 2), device="cuda")
ops.hash_scatter(T.g(table.detach()), tg, L, log2T, res.tolist(), T.g(o0), T.g(d0), T.g(e), T.g(g_enc), None, None)
got = tg.cpu(); ref = table.grad
bad = (got == 0) != (ref == 0)
print("mismatches", int(bad.sum()), "of", bad.numel())
ii = bad.nonzero()
for r, c in ii[:20].tolist():
    print("level", r >> log2T, "slot", r & (2**log2T - 1), "feat", c, "got", float(got[r, c]), "ref", float(ref[r, c]))
print("max abs diff", float((got - ref).abs().max()), "scale", float(ref.abs().max()))
