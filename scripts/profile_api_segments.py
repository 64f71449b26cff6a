"""Host time of the drop-in path's step, segment by segment (perf_counter around the statements of bench.one_step_api, no device sync inside
the loop, garbage collector frozen as in bench.py): where the ~1 ms of Python per step goes.  AMP=1: the reference Trainer's autocast +
torch.amp.GradScaler sequence (engine/trainer.py:470-495) with its two get_scale() host synchronisations."""
import functools, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc
from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers
from nerfstudio_thermal_amd.rays import RayBundle
dev = torch.device("cuda", 0)
cfg, arena, model = bench.build_model(dev)
opt = Optimizers(model.get_param_groups(), optimizer_cls=HipFusedAdam)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
dm = bench._datamanager(model, cam_t, cache, 4096)
cbs = model.get_training_callbacks()
groups = list(opt.optimizers.keys())
AMP = os.environ.get("AMP", "0") == "1"
scaler = torch.amp.GradScaler("cuda") if AMP else None
acc = {}
def seg(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
def step(s, timed):
    t = time.perf_counter()
    o, d, cam, im, th = dm.next_train(s); t = seg("next_train", t) if timed else t
    for cb in cbs: cb.run_callback_at_location(s, Loc.BEFORE_TRAIN_ITERATION)
    opt.zero_grad_some(groups); t = seg("callbacks + zero_grad", t) if timed else t
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=cam[:, None])
    batch = {"image": im, "is_thermal": th}; t = seg("RayBundle", t) if timed else t
    with torch.autocast(device_type="cuda", enabled=AMP):
        out = model(rb); t = seg("model(rb)", t) if timed else t
        m = model.get_metrics_dict(out, batch); t = seg("get_metrics_dict", t) if timed else t
        L = model.get_loss_dict(out, batch, m); t = seg("get_loss_dict", t) if timed else t
        loss = functools.reduce(torch.add, L.values()); t = seg("sum of losses", t) if timed else t
    if not AMP:
        loss.backward(); t = seg("backward", t) if timed else t
        opt.optimizer_step_all(s); t = seg("optimizer_step_all", t) if timed else t
        opt.scheduler_step_all(s); t = seg("scheduler_step_all", t) if timed else t
    else:
        sl = scaler.scale(loss); t = seg("scaler.scale(loss)", t) if timed else t
        sl.backward(); t = seg("backward", t) if timed else t
        opt.optimizer_scaler_step_some(scaler, groups); t = seg("optimizer_scaler_step_some", t) if timed else t
        sc = scaler.get_scale(); t = seg("get_scale() #1 (host sync: waits for the step's kernels)", t) if timed else t
        scaler.update(); t = seg("scaler.update()", t) if timed else t
        if sc <= scaler.get_scale():
            t = seg("get_scale() #2", t) if timed else t
            opt.scheduler_step_all(s); t = seg("scheduler_step_all", t) if timed else t
    for cb in cbs: cb.run_callback_at_location(s, Loc.AFTER_TRAIN_ITERATION)
    seg("after callbacks", t) if timed else None
s = 0
for _ in range(40): step(s, False); s += 1
gc.collect(); gc.freeze(); torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step(s, True); s += 1
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"{n} steps: host enqueue {t_enq/n*1e3:.3f} ms/step, wall {t_all/n*1e3:.3f} ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:28s} {v/n*1e6:8.1f} us/step")
# ---- inside backward(): the two custom nodes run on the autograd engine's worker thread
from nerfstudio_thermal_amd import autograd_ops as F
from nerfstudio_thermal_amd import model as M
inner = {}
def wrap(cls, name):
    orig = cls.backward
    def timed(ctx, *a):
        t0 = time.perf_counter(); r = orig(ctx, *a); inner[name] = inner.get(name, 0.0) + time.perf_counter() - t0; return r
    cls.backward = staticmethod(timed)
wrap(F.TrainLosses, "TrainLosses.backward")
wrap(M._RenderFn, "_RenderFn.backward")
acc.clear()
t0 = time.perf_counter()
for _ in range(n): step(s, True); s += 1
torch.cuda.synchronize()
print(f"second pass: backward {acc['backward']/n*1e6:.1f} us/step, of which")
for k, v in inner.items(): print(f"  {k:28s} {v/n*1e6:8.1f} us/step")
