"""Where does get_metrics_dict spend its host time in the un-profiled trainer loop? (wrappers with perf_counter around the pieces)"""
import functools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd import ops, autograd_ops as F
from nerfstudio_thermal_amd.optim import Optimizers
dev = torch.device("cuda", 0)
cfg, arena, model = bench.build_model(dev)
opt = Optimizers(model.get_param_groups())
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
T = {}
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T[label or name] = T.get(label or name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)
for n in ("pixel_losses", "proposal_losses", "camera_reg", "train_metrics"):
    wrap(ops, n)
wrap(model, "_loss_terms")
wrap(model, "get_metrics_dict")
wrap(model, "get_loss_dict")
orig_fwd = F.TrainLosses.forward
def fwd(ctx, *a):
    t0 = time.perf_counter(); r = orig_fwd(ctx, *a); T["TrainLosses.forward"] = T.get("TrainLosses.forward", 0.0) + time.perf_counter() - t0; return r
F.TrainLosses.forward = staticmethod(fwd)
for step in range(160):
    if step == 60:
        torch.cuda.synchronize(); T.clear(); t0 = time.perf_counter()
    bench.one_step_api(model, opt, cam_t, cache, 4096, step)
torch.cuda.synchronize()
print(f"wall {(time.perf_counter()-t0)/100*1e3:.3f} ms/step")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v/100*1e3:7.3f} ms/step")
