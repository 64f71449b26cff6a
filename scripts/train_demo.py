"""End-to-end demonstration on the synthetic RGB+T scene of bench.py: train BASELINE config 1 for a few thousand fused steps (new pixels every
step), then render every camera in full through the eval path and report PSNR against the synthetic images per spectrum.
(The targets are view-dependent analytic images, not a consistent 3-D scene: the point is that the whole loop -- sampling, raygen, forward,
losses, backward, Adam, schedules, eval render -- runs at full size and converges.  The targets differ per camera by construction, so the
model stores that in the per-camera appearance embedding: eval mode uses the MEAN embedding (use_average_appearance_embedding, as the
reference does) and its PSNR against these targets is low by design; the fit is reported on fresh pixels in training mode.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nerfstudio_thermal_amd import ops, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, _, _, _ = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
cams = synth.synth_cameras()
imgs = synth.synth_images(cams)
log = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for step in range(steps):
    losses = bench.one_step(eng, cam_t, cache, 4096, step, None)
    if step % 250 == 0 or step == steps - 1:
        torch.cuda.synchronize()
        log.append({"step": step, "seconds": time.perf_counter() - t0, **{k: float(v) for k, v in losses.items()}})
torch.cuda.synchronize(); train_s = time.perf_counter() - t0
# fresh pixels, training-mode render (per-camera appearance embedding + pose correction, as during training)
mse = {"rgb": [], "thermal": []}
for _ in range(10):
    u = torch.rand((1024, 3), device=dev)
    idx, img, is_th, cam = ops.sample_pixels(cache, 4096, u, 2, want_camera_indices=True)
    o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
    out, _ = eng.get_outputs(o, d, cam, training=True)
    m = is_th > 0
    mse["rgb"].append(float(((out["rgb"] - img)[~m] ** 2).mean()))
    mse["thermal"].append(float(((out["rgb_thermal"][:, 0] - img[:, 0])[m] ** 2).mean()))
fresh_psnr = {k: float(-10 * np.log10(np.mean(v))) for k, v in mse.items()}
psnr = {}
t1 = time.perf_counter()
for c in range(len(imgs)):
    H, W = imgs[c].shape[:2]
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    idx = torch.from_numpy(np.stack([np.full(H * W, c), yy.reshape(-1), xx.reshape(-1)], 1).astype(np.int64)).to(dev)
    outs = []
    for s in range(0, idx.shape[0], 32768):
        ii = idx[s:s + 32768]
        o, d, _, _ = ops.raygen(ii, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
        out, _ = eng.get_outputs(o, d, ii[:, 0].contiguous(), training=False)
        outs.append(out["rgb_thermal"] if cams["is_thermal"][c] else out["rgb"])
    pred = torch.cat(outs).cpu().numpy()
    gt = imgs[c].reshape(H * W, 3)[:, :1] if cams["is_thermal"][c] else imgs[c].reshape(H * W, 3)
    psnr[f"camera{c}_{'thermal' if cams['is_thermal'][c] else 'rgb'}"] = float(-10 * np.log10(((pred - gt) ** 2).mean()))
torch.cuda.synchronize(); eval_s = time.perf_counter() - t1
print(json.dumps({"steps": steps, "train_seconds": train_s, "train_rays_per_s": steps * 4096 / train_s, "psnr_db_fresh_pixels_train_mode": fresh_psnr, "eval_seconds_8_images": eval_s,
                  "psnr_db_eval_mode_mean_appearance": psnr, "curve": log}))
