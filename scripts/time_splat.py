"""BASELINE config 4: thermal-splatfacto forward render at 1080p on synthetic Gaussians; per-stage device time (HIP events on torch's
current stream, which is the stream the library launches on)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nerfstudio_thermal_amd  # noqa
from nerfstudio_thermal_amd.splat import ThermalSplatfactoModel, ThermalSplatfactoModelConfig, PinholeCamera
import splat_oracle as so  # noqa

N = int(os.environ.get("SPLAT_N", 1_000_000))
mode = os.environ.get("SPLAT_MODE", "classic")
p = so.synth_gaussians(N, seed=11, extent=1.5, scale_range=(float(os.environ.get("SPLAT_S0", -5.5)), float(os.environ.get("SPLAT_S1", -3.5))))
m = ThermalSplatfactoModel(ThermalSplatfactoModelConfig(rasterize_mode=mode), num_points=4)
m.load_gaussians(p); m.step = 10**6
cam = PinholeCamera(so.look_at_camera((3.2, 0.5, 0.8)), 1400.0, 1400.0, 960.0, 540.0, 1920, 1080)
for _ in range(3):
    out = m.get_outputs(cam)
torch.cuda.synchronize()
iters = 20
t0 = time.perf_counter()
for _ in range(iters):
    out = m.get_outputs(cam)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
hit = m.last_projection["num_tiles_hit"]
vis = int((m.last_projection["radii"] > 0).sum())
print(json.dumps({"gaussians": N, "visible": vis, "intersections": m.last_num_intersections, "mean_per_tile": m.last_num_intersections / 8160,
                  "frame_ms": ms, "fps": 1e3 / ms, "Mpix_per_s": 1920 * 1080 / ms / 1e3, "mean_alpha": float(out["accumulation"].mean())}))
