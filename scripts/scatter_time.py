"""Times the three table scatters (bin + fold) on the bench workload; with rocprofv3 around it the trace splits the two kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfstudio_thermal_amd import _lib
if os.environ.get("TN_LIB"):  # A/B timing against another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["TN_LIB"])
import bench
from nerfstudio_thermal_amd import ops
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
out, br = eng.get_outputs(o, d, idx[:, 0].contiguous(), True)
b = br[""]; lv = b.levels
which = sys.argv[1:] or ["prop0", "prop1", "main"]
d_o, d_d = torch.zeros_like(o), torch.zeros_like(d)
for name, net, L in (("prop0", eng.props[0], lv[0]), ("prop1", eng.props[1], lv[1]), ("main", eng.field, lv[2])):
    if name not in which:
        continue
    N, S = L.e_bins.shape[0], L.e_bins.shape[1] - 1
    # level-major [L][P] float2, as the backward kernels of all three grids hand their d enc over
    g_enc = (torch.randn((N * S, net.num_levels, 2), device=dev) * 1e-3).permute(1, 0, 2).contiguous()
    ms = bench.time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions, L.e_bins, g_enc, None, None))
    ms2 = bench.time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions, L.e_bins, g_enc, d_o, d_d))
    print(f"{name}: {ms*1e3:.1f} us without d position, {ms2*1e3:.1f} us with   [{os.environ.get('TN_FOLD_DBG','0')}]")
