"""Data-parallel schedule on a 1-rank group, NO per-step synchronisation (as bench.py runs it): host time of every step, outliers listed."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench
from nerfstudio_thermal_amd.parallel import OverlappedGradReducer, free_port
dev = torch.device("cuda", 0)
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
dist.init_process_group("nccl", rank=0, world_size=1)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
hook = OverlappedGradReducer(1)
for step in range(20):
    bench.one_step(eng, cam_t, cache, 4096, step, hook)
gc.collect(); gc.freeze()
torch.cuda.synchronize()
T = []
t00 = time.perf_counter()
for step in range(20, 120):
    t0 = time.perf_counter()
    bench.one_step(eng, cam_t, cache, 4096, step, hook)
    T.append((step, time.perf_counter() - t0))
torch.cuda.synchronize()
print(f"mean {(time.perf_counter() - t00) / 100 * 1e3:.3f} ms/step; host-only mean {sum(t for _, t in T) / 100 * 1e3:.3f}")
ts = sorted(t for _, t in T)
print("host per step: median %.3f ms, p90 %.3f, max %.3f" % (ts[50] * 1e3, ts[90] * 1e3, ts[-1] * 1e3))
print("outliers:", [(s, round(t * 1e3, 2)) for s, t in T if t > 3e-3][:20])
print("reserved MiB", torch.cuda.memory_reserved() / 2**20, "allocated", torch.cuda.memory_allocated() / 2**20)
st = torch.cuda.memory_stats()
print({k: st[k] for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "num_sync_all_streams") if k in st})
