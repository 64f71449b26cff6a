"""Scatter kernel on spatially random samples (no merging, no hot cells): does k_grid_scatter reach the request roof when the data does not
concentrate?  requests ~ 4.5 per sample per level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfstudio_thermal_amd import ops
dev = torch.device("cuda", 0)
cfg, arena, eng = bench.build_engine(dev)
torch.manual_seed(0)
for name, net, N, S in (("prop0", eng.props[0], 4096, 256), ("prop1", eng.props[1], 4096, 96), ("main", eng.field, 4096, 48)):
    # every sample its own random position: N*S "rays" with one sample each would change the launch shape, so keep [N,S] and randomise the
    # per-ray direction strongly + bins uniformly in [0, 2]: consecutive samples of a ray are ~2/S apart (>> cell size at fine levels)
    o = (torch.rand((N, 3), device=dev) * 2 - 1).contiguous()
    d = torch.nn.functional.normalize(torch.randn((N, 3), device=dev), dim=1).contiguous()
    e = torch.sort(torch.rand((N, S + 1), device=dev) * 2.0, dim=1)[0].contiguous()
    g_enc = torch.randn((N * S, 16 if net.num_levels == 5 else 32), device=dev) * 1e-3
    for ws in (True, False):
        ms = bench.time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, o, d, e, g_enc, None, None,
                                                    use_workspace=ws))
        req = N * S * net.num_levels * 4.5
        print(f"{name}: replicas={ws}: {ms*1e3:.1f} us for ~{req/1e6:.1f} M requests (upper estimate) -> {req/ms/1e6:.1f} G req/s")
