"""Timing probe for the backward ops (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfstudio_thermal_amd import _lib
if os.environ.get("TN_LIB"):  # A/B timing against another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["TN_LIB"])
import bench
from nerfstudio_thermal_amd import ops
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
cam = idx[:, 0].contiguous()
out, br = eng.get_outputs(o, d, cam, True)
b = br[""]; lv = b.levels
d_o, d_d = torch.zeros_like(o), torch.zeros_like(d)
for i in range(2):
    g = torch.rand_like(lv[i].density)
    t1 = bench.time_ms(lambda: ops.prop_density_bwd(eng.props[i], b.origins, b.directions, lv[i].e_bins, g, d_o, d_d))
    t2 = bench.time_ms(lambda: ops.prop_density_bwd(eng.props[i], b.origins, b.directions, lv[i].e_bins, g, None, None))
    print(f"prop bwd level {i}: with dpos {t1*1e3:.0f} us, without {t2*1e3:.0f} us")
gd = torch.rand_like(lv[2].density); gc = torch.rand_like(b.rgb_samples)
t1 = bench.time_ms(lambda: ops.field_bwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, d_o, d_d))
t2 = bench.time_ms(lambda: ops.field_bwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, None, None))
print(f"field bwd: with dpos {t1*1e3:.0f} us, without {t2*1e3:.0f} us")
t = bench.time_ms(lambda: ops.field_fwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, True))
print(f"field fwd (pack+encode+mlp): {t*1e3:.0f} us")
