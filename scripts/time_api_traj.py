"""Wall time per step of the model-API path over many steps (50-step means): is there a warm-up transient?"""
import os, sys, time
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
torch.cuda.synchronize()
t0 = time.perf_counter()
for step in range(800):
    bench.one_step_api(model, opt, cam_t, cache, 4096, step)
    if step % 50 == 49:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"steps {step-49:4d}-{step:4d}: {(t1-t0)/50*1e3:.3f} ms/step  allocated {torch.cuda.memory_allocated()/2**20:.0f} MiB reserved {torch.cuda.memory_reserved()/2**20:.0f} MiB")
        t0 = time.perf_counter()
