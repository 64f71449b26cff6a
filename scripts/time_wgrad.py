"""field backward alone (MLP bwd + wgrad + scatter) for a sweep of TN_WGRAD_BLOCKS; run under rocprofv3 to split the kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd import ops
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
cam = idx[:, 0].contiguous()
out, br = eng.get_outputs(o, d, cam, True)
b = br[""]; lv = b.levels
gd = torch.rand_like(lv[2].density); gc = torch.rand_like(b.rgb_samples)
t1 = bench.time_ms(lambda: ops.field_bwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, None, None))
print(f"field bwd without dpos {t1*1e3:.0f} us  [TN_WGRAD_BLOCKS={os.environ.get('TN_WGRAD_BLOCKS')}]")
ph = ops._lib
def mlp_only():
    ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, None, None, ph.TN_BWD_MLP | ph.TN_BWD_JOIN)
t2 = bench.time_ms(mlp_only)
print(f"field bwd MLP + weight gradients only (no scatter) {t2*1e3:.0f} us")
