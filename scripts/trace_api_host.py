"""Inclusive host time per step of the functions under the drop-in path's iteration (perf_counter wrappers, no device sync): which Python
calls sit between the start of an iteration and the two library calls the GPU waits for (tn_render_rays_train, tn_render_rays_train_bwd).
AMP=1 (default): the reference Trainer's autocast + GradScaler sequence."""
import functools, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import nerfstudio_thermal_amd.ops as ops
import nerfstudio_thermal_amd.model as M
import nerfstudio_thermal_amd.engine as E
import nerfstudio_thermal_amd.autograd_ops as F
import nerfstudio_thermal_amd.optim as O
import nerfstudio_thermal_amd.model_components as MC
import nerfstudio_thermal_amd.data as D
acc, cnt = {}, {}
MARK = {"C:tn_sample_rays", "C:tn_render_rays_train", "C:tn_train_losses", "C:tn_render_rays_train_bwd", "C:tn_adam_step_ranges_amp"}
marks = []  # (name, t_begin, t_end) of the library calls the GPU waits for, per step
def wrap(obj, name, label=None, static=False):
    orig = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj)}.{name}"
    @functools.wraps(orig)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; cnt[label] = cnt.get(label, 0) + 1
    setattr(obj, name, staticmethod(timed) if static else timed)
lib = ops._lib.load()
class LibProxy:  # time the C calls themselves
    def __init__(self, lib): self._lib = lib; self._cache = {}
    def __getattr__(self, n):
        f = self._cache.get(n)
        if f is None:
            g = getattr(self._lib, n)
            if not callable(g): return g
            def f(*a, _g=g, _n="C:" + n):
                t0 = time.perf_counter(); r = _g(*a); t1 = time.perf_counter(); acc[_n] = acc.get(_n, 0.0) + t1 - t0; cnt[_n] = cnt.get(_n, 0) + 1
                if _n in MARK: marks.append((_n, t0, t1))
                return r
            self._cache[n] = f
        return f
proxy = LibProxy(lib)
ops._lib.load = lambda: proxy
for name in ("render_rays_train", "render_rays_train_bwd", "train_losses", "losses_finish", "train_metrics", "pose_apply_bwd", "adam_step_ranges_amp",
             "grad_nonfinite_ranges", "sample_rays"):
    if hasattr(ops, name): wrap(ops, name, "ops." + name)
wrap(M._RenderFn, "forward", "_RenderFn.forward", static=True)
wrap(M._RenderFn, "backward", "_RenderFn.backward", static=True)
wrap(F.TrainLosses, "forward", "TrainLosses.forward", static=True)
wrap(F.TrainLosses, "backward", "TrainLosses.backward", static=True)
wrap(E.RenderEngine, "get_outputs", "engine.get_outputs")
wrap(E.RenderEngine, "render_branch", "engine.render_branch")
wrap(M.ThermalNerfactoModel, "get_outputs", "model.get_outputs")
wrap(M.ThermalNerfactoModel, "_grads_alias_arena", "model._grads_alias_arena")
wrap(M.ThermalNerfactoModel, "_loss_terms", "model._loss_terms")
wrap(M.ThermalNerfactoModel, "get_metrics_dict", "model.get_metrics_dict")
wrap(O.Optimizers, "optimizer_scaler_step_some", "Optimizers.optimizer_scaler_step_some")
wrap(O.HipFusedAdam, "collect_runs", "HipFusedAdam.collect_runs")
AMP = os.environ.get("AMP", "1") == "1"
if os.environ.get("ST_BWD", "0") == "1":
    torch.autograd.set_multithreading_enabled(False)  # backward nodes run on the calling thread: no hand-over to the device thread
dev = torch.device("cuda", 0)
cfg, arena, model = bench.build_model(dev)
wrap(type(model.collider), "forward", "collider.forward")
opt = O.Optimizers(model.get_param_groups(), optimizer_cls=O.HipFusedAdam)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
scaler = torch.amp.GradScaler("cuda") if AMP else None
s = 0
for _ in range(40): bench.one_step_api(model, opt, cam_t, cache, 4096, s, scaler); s += 1
gc.collect(); gc.freeze(); torch.cuda.synchronize(); acc.clear(); cnt.clear()
n = 200
item = torch.Tensor.item
def timed_item(self):
    t0 = time.perf_counter(); r = item(self); marks.append(("item()", t0, time.perf_counter())); return r
torch.Tensor.item = timed_item
tl = {}
t0 = time.perf_counter()
for _ in range(n):
    marks.clear(); ts = time.perf_counter()
    bench.one_step_api(model, opt, cam_t, cache, 4096, s, scaler); s += 1
    te = time.perf_counter()
    seen = {}
    for name, a, b in marks:
        k = seen[name] = seen.get(name, 0) + 1
        key = f"{name} #{k}"
        e = tl.setdefault(key, [0.0, 0.0, 0]); e[0] += a - ts; e[1] += b - ts; e[2] += 1
    e = tl.setdefault("step end", [0.0, 0.0, 0]); e[0] += te - ts; e[1] += te - ts; e[2] += 1
torch.cuda.synchronize(); wall = time.perf_counter() - t0
torch.Tensor.item = item
print("host timeline of one iteration (mean begin / end of each call, us after the iteration's start; n = calls seen):")
for k, (a, b, c) in sorted(tl.items(), key=lambda kv: kv[1][0] / kv[1][2]): print(f"  {k:36s} {a/c*1e6:8.1f} -> {b/c*1e6:8.1f}   n={c}")
print(f"{n} steps, wall {wall/n*1e3:.3f} ms/step (with the wrappers' own cost)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:44s} {v/n*1e6:8.1f} us/step  ({cnt[k]/n:.1f} calls)")
