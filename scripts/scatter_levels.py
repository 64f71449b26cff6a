"""Per-level timing of k_grid_scatter (run on the GPU box): which levels of which grid are far from the atomic-request floor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfstudio_thermal_amd import _lib
if os.environ.get("TN_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TN_LIB"])
import bench
from nerfstudio_thermal_amd import ops
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
cam = idx[:, 0].contiguous()
out, br = eng.get_outputs(o, d, cam, True)
b = br[""]; lv = b.levels
grids = [("prop0", eng.props[0], lv[0]), ("prop1", eng.props[1], lv[1]), ("main", eng.field, lv[2])]
for name, net, L in grids:
    T = 2 ** net.log2_hashmap_size
    N, S = L.e_bins.shape[0], L.e_bins.shape[1] - 1
    tot = 0.0
    for l in range(net.num_levels):
        g_enc = torch.randn((N * S, 2), device=dev) * 1e-3
        tab = net.table[l * T:(l + 1) * T]
        grad = net.grads["table"][l * T:(l + 1) * T]
        ms = bench.time_ms(lambda: ops.hash_scatter(tab, grad, 1, net.log2_hashmap_size, [net.res[l]], b.origins, b.directions, L.e_bins, g_enc, None, None))
        tot += ms
        print(f"{name} level {l:2d} res {net.res[l]:7.1f}: {ms*1e3:7.1f} us  ({N*S/ms/1e6:.1f} G samples/s)")
    g_enc = torch.randn((N * S, 16 if net.num_levels == 5 else 32), device=dev) * 1e-3
    ms = bench.time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions,
                                                L.e_bins, g_enc, None, None))
    print(f"{name}: sum of single-level launches {tot*1e3:.1f} us, all levels in one launch {ms*1e3:.1f} us")

# no-atomics floor: an all-zero g_enc makes every lane skip its atomics (v != 0 is false), leaving loads + index math + scans
for name, net, L in grids:
    N, S = L.e_bins.shape[0], L.e_bins.shape[1] - 1
    g_enc = torch.zeros((N * S, 16 if net.num_levels == 5 else 32), device=dev)
    ms = bench.time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions,
                                                L.e_bins, g_enc, None, None))
    ms2 = bench.time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions,
                                                 L.e_bins, g_enc, None, None, use_workspace=False))
    print(f"{name}: zero-gradient launch (no atomics) {ms*1e3:.1f} us; without replica scratch {ms2*1e3:.1f} us")
