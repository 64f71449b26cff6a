"""Launch the three table scatters (random gradient, replica scratch on) a few times: target for PMC passes on the atomic path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd import ops
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
out, br = eng.get_outputs(o, d, idx[:, 0].contiguous(), True)
b = br[""]; lv = b.levels
for name, net, L in (("prop0", eng.props[0], lv[0]), ("prop1", eng.props[1], lv[1]), ("main", eng.field, lv[2])):
    N, S = L.e_bins.shape[0], L.e_bins.shape[1] - 1
    g_enc = torch.randn((N * S, 16 if net.num_levels == 5 else 32), device=dev) * 1e-3
    for _ in range(3):
        ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions, L.e_bins, g_enc, None, None)
torch.cuda.synchronize()
