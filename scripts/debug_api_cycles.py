"""Which objects of one model-API iteration are only freed by the cyclic collector? (each one delays the release of its device tensors)"""
import collections, gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd.optim import Optimizers
dev = torch.device("cuda", 0)
cfg, arena, model = bench.build_model(dev)
opt = Optimizers(model.get_param_groups())
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
for step in range(12):
    bench.one_step_api(model, opt, cam_t, cache, 4096, step)
gc.collect()
gc.disable()
a0 = torch.cuda.memory_allocated()
for step in range(12, 14):
    bench.one_step_api(model, opt, cam_t, cache, 4096, step)
torch.cuda.synchronize()
a1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
a2 = torch.cuda.memory_allocated()
print(f"allocated before {a0/2**20:.1f} MiB, after 2 steps {a1/2**20:.1f}, after gc {a2/2**20:.1f}; collected {n}")
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(25))
for o in gc.garbage:
    if type(o).__name__ in ("dict",) and len(o) < 12:
        print("dict keys", list(o.keys())[:12])
    if isinstance(o, torch.Tensor):
        print("tensor", tuple(o.shape), o.grad_fn)
