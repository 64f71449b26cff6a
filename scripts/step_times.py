"""Per-step device and host times of the first steps of a fresh process (what `bench.py --steps 20 --warmup 5` times): HIP events between the
steps on the launch stream, perf_counter around each enqueue, and the allocator's segment count (a growing pool = hipMalloc inside a step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfstudio_thermal_amd import _lib
if os.environ.get("TN_LIB"):  # A/B timing against another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["TN_LIB"])
import bench
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
from nerfstudio_thermal_amd.optim import DeviceGradScaler
scaler = DeviceGradScaler(dev, num_groups=len(arena.optimised_groups))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host, segs, upd = [], [], []
torch.cuda.synchronize()
evs[0].record()
for s in range(n):
    t0 = time.perf_counter()
    bench.one_step(eng, cam_t, cache, 4096, s, None, scaler)
    host.append((time.perf_counter() - t0) * 1e3)
    upd.append(int(eng.steps_since_update == 1))
    segs.append(torch.cuda.memory_stats()["segment.all.current"])
    evs[s + 1].record()
    if os.environ.get("SYNC_EACH") == "1":
        torch.cuda.synchronize()
torch.cuda.synchronize()
for s in range(n):
    print(f"step {s:3d} upd {upd[s]} device {evs[s].elapsed_time(evs[s+1]):7.3f} ms  host enqueue {host[s]:7.3f} ms  segments {segs[s]}")

dev_ms = np.array([evs[s].elapsed_time(evs[s + 1]) for s in range(n)])
u = np.array(upd, dtype=bool)
skip = 12  # every step below 10 updates the proposal networks, and the first steps allocate
print(f"[{os.environ.get('TN_LIB', 'default lib')}] median device ms: update steps {np.median(dev_ms[skip:][u[skip:]]):.4f}  other steps {np.median(dev_ms[skip:][~u[skip:]]):.4f}  all {np.median(dev_ms[skip:]):.4f}")
