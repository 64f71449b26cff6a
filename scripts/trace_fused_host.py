"""Inclusive host time per fused step (RenderEngine.train_step) of the library calls and the Python around them (perf_counter wrappers, no
device sync inside the loop): what a single-call tn_train_step could remove."""
import functools, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import nerfstudio_thermal_amd.ops as ops
import nerfstudio_thermal_amd.engine as E
from nerfstudio_thermal_amd.optim import DeviceGradScaler
acc, cnt = {}, {}
def wrap(obj, name, label):
    orig = getattr(obj, name)
    @functools.wraps(orig)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; cnt[label] = cnt.get(label, 0) + 1
    setattr(obj, name, timed)
lib = ops._lib.load()
class LibProxy:
    def __init__(self, lib): self._lib = lib; self._cache = {}
    def __getattr__(self, n):
        f = self._cache.get(n)
        if f is None:
            g = getattr(self._lib, n)
            if not callable(g): return g
            def f(*a, _g=g, _n="C:" + n):
                t0 = time.perf_counter(); r = _g(*a); acc[_n] = acc.get(_n, 0.0) + time.perf_counter() - t0; cnt[_n] = cnt.get(_n, 0) + 1; return r
            self._cache[n] = f
        return f
ops._lib.load = lambda p=LibProxy(lib): p
for name in ("render_rays_train", "render_rays_train_bwd", "train_losses", "losses_finish", "pose_bwd_finish", "adam_step_ranges_amp", "sample_rays"):
    wrap(ops, name, "ops." + name)
for name in ("get_outputs", "loss_and_backward", "optimizer_step", "train_step"):
    wrap(E.RenderEngine, name, "engine." + name)
rays = int(os.environ.get("RAYS", "4096"))
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, rays, 42)
cache = bench.make_image_cache(dev)
scaler = DeviceGradScaler(dev, num_groups=len(arena.optimised_groups))
s = 0
for _ in range(60): bench.one_step(eng, cam_t, cache, rays, s, None, scaler); s += 1
gc.collect(); gc.freeze(); torch.cuda.synchronize(); acc.clear(); cnt.clear()
n = 200
t0 = time.perf_counter()
for _ in range(n): bench.one_step(eng, cam_t, cache, rays, s, None, scaler); s += 1
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"{rays} rays, {n} steps: host enqueue {t_enq/n*1e3:.3f} ms/step, wall {wall/n*1e3:.3f} ms/step (with the wrappers' own cost)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:40s} {v/n*1e6:8.1f} us/step  ({cnt[k]/n:.1f} calls)")
