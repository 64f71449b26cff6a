"""Host-side profile (cProfile) of the drop-in path's step: where the Python time of the reference Trainer's sequence goes."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers
dev = torch.device("cuda", 0)
cfg, arena, model = bench.build_model(dev)
opt = Optimizers(model.get_param_groups(), optimizer_cls=HipFusedAdam)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
scaler = torch.amp.GradScaler("cuda") if "--scaler" in sys.argv else None
step = 0
for _ in range(30):
    bench.one_step_api(model, opt, cam_t, cache, 4096, step, scaler); step += 1
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(100):
    bench.one_step_api(model, opt, cam_t, cache, 4096, step, scaler); step += 1
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"100 steps: host enqueue {t_enq*10:.3f} ms/step, wall {t_all*10:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    bench.one_step_api(model, opt, cam_t, cache, 4096, step, scaler); step += 1
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:8000])
