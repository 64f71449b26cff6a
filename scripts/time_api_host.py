"""Host-side cost of the model-API step, phase by phase (no device synchronisation inside the loop: what the Python thread spends enqueueing)."""
import functools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc
from nerfstudio_thermal_amd.optim import Optimizers
from nerfstudio_thermal_amd.rays import RayBundle
dev = torch.device("cuda", 0)
cfg, arena, model = bench.build_model(dev)
opt = Optimizers(model.get_param_groups())
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
dm = bench._datamanager(model, cam_t, cache, 4096)
cbs = model.get_training_callbacks()
T = {}
def tick(name, t0):
    t = time.perf_counter(); T[name] = T.get(name, 0.0) + (t - t0); return t
for step in range(120):
    if step == 20:
        torch.cuda.synchronize(); T.clear(); wall0 = time.perf_counter()
    t = time.perf_counter()
    o, d, cam, im, th = dm.next_train(step); t = tick("data", t)
    for cb in cbs: cb.run_callback_at_location(step, Loc.BEFORE_TRAIN_ITERATION)
    opt.zero_grad_all(); t = tick("cb+zero_grad", t)
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=cam[:, None])
    out = model(rb); t = tick("forward", t)
    batch = {"image": im, "is_thermal": th}
    m = model.get_metrics_dict(out, batch); t = tick("metrics", t)
    ld = model.get_loss_dict(out, batch, m); t = tick("loss_dict", t)
    tot = functools.reduce(torch.add, ld.values()); t = tick("sum", t)
    tot.backward(); t = tick("backward", t)
    opt.optimizer_step_all(step); t = tick("optim", t)
    opt.scheduler_step_all(step)
    for cb in cbs: cb.run_callback_at_location(step, Loc.AFTER_TRAIN_ITERATION)
    t = tick("sched+cb", t)
host = time.perf_counter() - wall0
torch.cuda.synchronize()
wall = time.perf_counter() - wall0
print(f"host enqueue {host/100*1e3:.3f} ms/step, wall {wall/100*1e3:.3f} ms/step")
for k, v in T.items():
    print(f"  {k:14s} {v/100*1e3:7.3f} ms")
# ---- finer: where does forward spend its host time?
import cProfile, pstats
pr = cProfile.Profile()
for step in range(120, 170):
    o, d, cam, im, th = dm.next_train(step)
    opt.zero_grad_all()
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=cam[:, None])
    out = model(rb)
    batch = {"image": im, "is_thermal": th}
    pr.enable(); md = model.get_metrics_dict(out, batch); pr.disable()
    ld = model.get_loss_dict(out, batch, md)
    functools.reduce(torch.add, ld.values()).backward()
    opt.optimizer_step_all(step)
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(30)
