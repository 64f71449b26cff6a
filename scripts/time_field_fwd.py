"""Times tn_field_fwd stand-alone on the bench workload (4096 rays x 48 samples): training and inference variant, plus the backward MLP phase
and k_field_dpos beside it.  TN_LIB=<variant build> for the FWD_ABLATE / layout A/Bs."""
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
tag = os.environ.get("TN_LIB", "default")
tt = bench.time_ms(lambda: ops.field_fwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, True), iters=20, warmup=3)
te = bench.time_ms(lambda: ops.field_fwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, False), iters=20, warmup=3)
print(f"[{tag}] field fwd (pack + fused launch): training {tt*1e3:.1f} us, inference {te*1e3:.1f} us")
if os.environ.get("TIME_BWD", "1") == "1":
    ops.field_fwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, True)
    gd = torch.rand_like(lv[2].density); gc = torch.rand_like(b.rgb_samples)
    d_o, d_d = torch.zeros_like(o), torch.zeros_like(d)
    t = bench.time_ms(lambda: ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, None, None, _lib.TN_BWD_MLP | _lib.TN_BWD_JOIN), iters=20, warmup=3)
    t2 = bench.time_ms(lambda: ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, d_o, d_d, _lib.TN_BWD_MLP | _lib.TN_BWD_JOIN), iters=20, warmup=3)
    print(f"[{tag}] field bwd MLP phase {t*1e3:.1f} us, with d position {t2*1e3:.1f} us")
