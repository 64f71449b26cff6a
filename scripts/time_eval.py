"""Eval-mode render throughput (run on the GPU box): one 640x480 RGB camera + one 160x120 thermal camera, 32768-ray chunks
(Model.get_outputs_for_camera_ray_bundle, models/base_model.py:177-205)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nerfstudio_thermal_amd import ops, synth
dev = torch.device("cuda", 0)
for mode in ("shared", "separate"):
    cfg, arena, eng = bench.build_engine(dev, mode=mode)
    cams = synth.synth_cameras()
    t = lambda k: torch.from_numpy(cams[k]).to(dev)
    for c in (0, 4):
        H, W = int(cams["height"][c]), int(cams["width"][c])
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        idx = torch.from_numpy(np.stack([np.full(H * W, c), yy.reshape(-1), xx.reshape(-1)], 1).astype(np.int64)).to(dev)
        def render():
            outs = []
            for s in range(0, idx.shape[0], 32768):
                ii = idx[s:s + 32768]
                o, d, _, _ = ops.raygen(ii, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
                out, _ = eng.get_outputs(o, d, ii[:, 0].contiguous(), training=False)
                outs.append(out["rgb"])
            return torch.cat(outs)
        render(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            render()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{mode}: camera {c} {W}x{H}: {dt*1e3:.2f} ms/image = {H*W/dt/1e6:.2f} M rays/s")
