"""Timing probe for the small per-ray kernels of a training step (run on the GPU box): back-to-back launches, HIP-event timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd import ops
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
N = int(os.environ.get("RAYS", "4096"))
cam_t, idx, img, is_th = bench.make_batch(dev, N, 42)
o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
cam = idx[:, 0].contiguous()
out, br = eng.get_outputs(o, d, cam, True)
b = br[""]; lv = b.levels
T = lambda fn: bench.time_ms(fn, iters=50, warmup=5) * 1e3
L = torch.zeros(16, device=dev); Lp = torch.zeros((ops.LOSS_LINES, 16), device=dev)
dws = [torch.zeros_like(l.weights) for l in lv]
dcomp = torch.zeros_like(b.comp)
props = [(lv[i].s_bins, lv[i].weights, dws[i]) for i in range(2)]
pixel = (b.comp[:, :3], b.comp[:, 3:], img, is_th, 100.0, 1e-6, 1e-6, dcomp[:, :3], dcomp[:, 3:])
print("render_fwd            %6.1f us" % T(lambda: ops.render_fwd(lv[2].e_bins, lv[2].density, b.rgb_samples, True)))
print("render_fwd (no depth) %6.1f us" % T(lambda: ops.render_fwd(lv[2].e_bins, lv[2].density, b.rgb_samples, True, want_depth=False)))
print("train_losses (all)    %6.1f us" % T(lambda: ops.train_losses(lv[2].s_bins, lv[2].weights, props, 0.002, 1.0, dws[2], Lp, pixel=pixel)))
print("train_losses no pixel %6.1f us" % T(lambda: ops.train_losses(lv[2].s_bins, lv[2].weights, props, 0.002, 1.0, dws[2], Lp)))
print("train_losses no grads %6.1f us" % T(lambda: ops.train_losses(lv[2].s_bins, lv[2].weights, [(p[0], p[1], None) for p in props], 0.002, 1.0, None, Lp)))
print("distortion            %6.1f us" % T(lambda: ops.distortion_loss(lv[2].s_bins, lv[2].weights, 0.002, L[9:10], dws[2])))
for i in range(2):
    print("interlevel vs level %d %6.1f us" % (i, T(lambda: ops.interlevel_loss(lv[2].s_bins, lv[2].weights, lv[i].s_bins, lv[i].weights, 1.0, L[8:9], dws[i]))))
    print("  without gradient    %6.1f us" % T(lambda: ops.interlevel_loss(lv[2].s_bins, lv[2].weights, lv[i].s_bins, lv[i].weights, 1.0, L[8:9], None)))
print("pixel_losses          %6.1f us" % T(lambda: ops.pixel_losses(*pixel[:7], L[0:8], *pixel[7:])))
print("losses_finish         %6.1f us" % T(lambda: ops.losses_finish(Lp, L)))
print("render_bwd            %6.1f us" % T(lambda: ops.render_bwd(lv[2].e_bins, lv[2].density, b.rgb_samples, lv[2].weights, dcomp, dws[2])))
jit = torch.rand(N, device=dev)
nears, fars = eng._nears_fars(N, True)
print("spaced_bins           %6.1f us" % T(lambda: ops.spaced_bins(nears, fars, 256, jit)))
print("weights_resample 256  %6.1f us" % T(lambda: ops.weights_resample(lv[0].e_bins, lv[0].density, lv[0].s_bins, 96, 1.0, nears, fars, jit)))
print("weights_resample 96   %6.1f us" % T(lambda: ops.weights_resample(lv[1].e_bins, lv[1].density, lv[1].s_bins, 48, 1.0, nears, fars, jit)))
print("empty torch op        %6.1f us" % T(lambda: L.zero_()))
