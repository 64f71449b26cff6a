"""Times the renderers / losses / renderer backward of the last level on the bench workload: the three entry points one by one and
tn_render_losses_bwd (all terms, without the interlevel slices, without proposal gradients)."""
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
out, br = eng.get_outputs(o, d, idx[:, 0].contiguous(), True)
b = br[""]; lv = b.levels
N = o.shape[0]
e2, dn2, s2, rgb = lv[2].e_bins, lv[2].density, lv[2].s_bins, b.rgb_samples
d0, d1, d2, dc = torch.zeros_like(lv[0].weights), torch.zeros_like(lv[1].weights), torch.zeros_like(lv[2].weights), torch.zeros((N, 4), device=dev)
Lp = torch.zeros((ops.LOSS_LINES, 16), device=dev)
w, c, a, m, x = ops.render_fwd(e2, dn2, rgb, True)
pix = (c[:, :3], c[:, 3:], img, is_th, 100.0, 1e-3, 1e-3, dc[:, :3], dc[:, 3:])
which = sys.argv[1:]  # e.g. "2": only the third variant (under rocprofv3 the kernel durations are the measurement: the host-side timing below
# is bound by the binding's argument checks, ~40 us per call, not by these kernels)
for vi, (name, props) in enumerate((("update step (d weights of both proposal levels)", [(lv[0].s_bins, lv[0].weights, d0), (lv[1].s_bins, lv[1].weights, d1)]),
                    ("other steps", [(lv[0].s_bins, lv[0].weights, None), (lv[1].s_bins, lv[1].weights, None)]), ("no interlevel slices", []))):
    if which and str(vi) not in which:
        continue
    t1 = bench.time_ms(lambda: ops.render_fwd(e2, dn2, rgb, True))
    t2 = bench.time_ms(lambda: ops.train_losses(s2, w, props, 0.002, 1.0, d2, Lp, pixel=pix))
    t3 = bench.time_ms(lambda: ops.render_bwd(e2, dn2, rgb, w, dc, d2))
    t4 = bench.time_ms(lambda: ops.render_losses_bwd(e2, dn2, rgb, s2, props, 0.002, 1.0, d2, img, is_th, 100.0, 1e-3, 1e-3, dc, Lp))
    t5 = bench.time_ms(lambda: ops.render_losses_bwd(e2, dn2, rgb, s2, props, 0.002, 1.0, d2, img, is_th, 100.0, 1e-3, 1e-3, dc, Lp, clip_depth=False))
    print(f"{name}: render_fwd {t1*1e3:.1f} + train_losses {t2*1e3:.1f} + render_bwd {t3*1e3:.1f} = {(t1+t2+t3)*1e3:.1f} us;  render_losses_bwd {t4*1e3:.1f} us, without the clip launch {t5*1e3:.1f}")
