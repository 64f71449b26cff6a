import functools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd import ops, engine as E, model as M
from nerfstudio_thermal_amd.optim import Optimizers
from nerfstudio_thermal_amd.rays import RayBundle
dev = torch.device("cuda", 0)
cfg, arena, model = bench.build_model(dev)
opt = Optimizers(model.get_param_groups())
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
dm = bench._datamanager(model, cam_t, cache, 4096)
T = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[label] = T.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for n in ("pose_apply_fwd", "spaced_bins", "weights_resample", "prop_density_fwd", "field_fwd", "weights_fwd", "composite_fwd", "field_pack"):
    wrap(ops, n, "ops." + n)
wrap(model.engine, "get_outputs", "engine.get_outputs")
wrap(model.arena, "zero_grad", "arena.zero_grad")
wrap(model, "_grads_alias_arena", "alias check")
for step in range(120):
    if step == 20:
        torch.cuda.synchronize(); T.clear()
    o, d, cam, im, th = dm.next_train(step)
    opt.zero_grad_all()
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=cam[:, None])
    t0 = time.perf_counter(); out = model(rb); T["forward total"] = T.get("forward total", 0.0) + time.perf_counter() - t0
    batch = {"image": im, "is_thermal": th}
    ld = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
    functools.reduce(torch.add, ld.values()).backward()
    opt.optimizer_step_all(step)
torch.cuda.synchronize()
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v/100*1e3:7.3f} ms")
# ---- what inside spaced_bins blocks?
import ctypes as C
from nerfstudio_thermal_amd import _lib
T.clear()
orig = ops.__dict__["spaced_bins"].__closure__[0].cell_contents  # unwrap
def spaced_bins(nears, fars, S, jitter=None):
    N = nears.shape[0]
    t0 = time.perf_counter()
    s = torch.empty((N, S + 1), device=nears.device); e = torch.empty((N, S + 1), device=nears.device)
    t1 = time.perf_counter()
    args = (ops._f32(ops._lin_table("spaced", S, nears.device), "lin"), ops._ray_scalar(jitter, "jitter", N, True), ops._ray_scalar(nears, "nears", N),
            ops._ray_scalar(fars, "fars", N), N, S, ops._f32(s, "s"), ops._f32(e, "e"), ops._stream())
    t2 = time.perf_counter()
    _lib.check(_lib.load().tn_spaced_bins(*args), "tn_spaced_bins")
    t3 = time.perf_counter()
    T["sb.alloc"] = T.get("sb.alloc", 0) + t1 - t0; T["sb.args"] = T.get("sb.args", 0) + t2 - t1; T["sb.launch"] = T.get("sb.launch", 0) + t3 - t2
    return s, e
ops.spaced_bins = spaced_bins
for step in range(120, 220):
    o, d, cam, im, th = dm.next_train(step)
    opt.zero_grad_all()
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=cam[:, None])
    t0 = time.perf_counter(); out = model(rb); T["forward total"] = T.get("forward total", 0.0) + time.perf_counter() - t0
    batch = {"image": im, "is_thermal": th}
    ld = model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))
    functools.reduce(torch.add, ld.values()).backward()
    opt.optimizer_step_all(step)
torch.cuda.synchronize()
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    if True: print(f"  {k:24s} {v/100*1e3:7.3f} ms")
