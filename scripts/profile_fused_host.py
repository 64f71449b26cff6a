"""Host-side profile of the fused step (cProfile over 100 steps, no device synchronisation inside): where the Python thread spends its time."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
cache = bench.make_image_cache(dev)
for step in range(30):
    bench.one_step(eng, cam_t, cache, 4096, step, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for step in range(30, 130):
    bench.one_step(eng, cam_t, cache, 4096, step, None)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"host enqueue {host/100*1e3:.3f} ms/step, wall {wall/100*1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for step in range(130, 230):
    bench.one_step(eng, cam_t, cache, 4096, step, None)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(28)
