"""Times the MLP phase of the main field's backward alone (tn_field_bwd_phase(TN_BWD_MLP)); TN_FB_ABLATE bits are diagnostics."""
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
gd = torch.rand_like(lv[2].density); gc = torch.rand_like(b.rgb_samples)
for abl in ["-"]:
    t = bench.time_ms(lambda: ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, None, None, _lib.TN_BWD_MLP | _lib.TN_BWD_JOIN), iters=20, warmup=3)
    print(f"field bwd MLP phase, fused={os.environ.get('TN_FIELD_BWD_FUSED','1')} lib={os.environ.get('TN_LIB','default')}: {t*1e3:.1f} us")
