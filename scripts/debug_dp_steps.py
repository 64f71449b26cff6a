"""Per-step wall time of the data-parallel schedule on a 1-rank RCCL group (synchronised after every step): outliers? which steps?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench
from nerfstudio_thermal_amd.parallel import OverlappedGradReducer, GradAllReducer, free_port
dev = torch.device("cuda", 0)
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
dist.init_process_group("nccl", rank=0, world_size=1)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
chunks = int(os.environ.get("CHUNKS", "-1"))
hook = OverlappedGradReducer(1) if chunks < 0 else OverlappedGradReducer(1, level_chunks=chunks)
import gc
if os.environ.get('FREEZE'):
    for step in range(5): bench.one_step(eng, cam_t, cache, 4096, step, hook)
    gc.collect(); gc.freeze()
T = []
gc.callbacks.append(lambda phase, info: print('gc', phase, info) if info.get('generation') == 2 else None)
for step in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.one_step(eng, cam_t, cache, 4096, step, hook)
    th = time.perf_counter()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    T.append((step, eng.last_updated, (th - t0) * 1e3, (t1 - t0) * 1e3))
for s, u, h, w in [t for t in T if t[3] > 3.0]:
    print(f"step {s:3d} update={int(bool(u))} host {h:6.3f} ms wall {w:6.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for step in range(60, 100):
    bench.one_step(eng, cam_t, cache, 4096, step, hook)
pr.disable(); torch.cuda.synchronize()
print('mean wall', sum(t[3] for t in T[10:]) / len(T[10:]))
