"""HIP-event timing of the proposal networks' entry points on the bench workload (TN_LIB=<other build> for A/B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfstudio_thermal_amd import _lib
if os.environ.get("TN_LIB"):
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
    tf = bench.time_ms(lambda: ops.prop_density_fwd(eng.props[i], b.origins, b.directions, lv[i].e_bins))
    tb = bench.time_ms(lambda: ops.prop_density_bwd(eng.props[i], b.origins, b.directions, lv[i].e_bins, g, d_o, d_d))
    print(f"prop level {i}: forward {tf*1e3:.1f} us, backward (mlp + scatter, with d position) {tb*1e3:.1f} us")
