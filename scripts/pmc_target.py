"""Target of the PMC passes (scripts/pmc_passes.sh): every hash-grid gather / scatter entry point and the MFMA kernels of the bench workload
(4096 rays, shared mode), each launched alone a few times so that per-launch counters are not mixed with concurrent streams."""
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
cam = idx[:, 0].contiguous()
out, br = eng.get_outputs(o, d, cam, True)
b = br[""]; lv = b.levels
d_o, d_d = torch.zeros_like(o), torch.zeros_like(d)
R = 3
for i in range(2):
    for _ in range(R):
        ops.prop_density_fwd(eng.props[i], b.origins, b.directions, lv[i].e_bins)
    torch.cuda.synchronize()
for _ in range(R):
    ops.field_fwd(eng.field, b.origins, b.directions, cam, lv[2].e_bins, True)
torch.cuda.synchronize()
for net, L in ((eng.props[0], lv[0]), (eng.props[1], lv[1]), (eng.field, lv[2])):
    N, S = L.e_bins.shape[0], L.e_bins.shape[1] - 1
    # level-major [L][P] float2, as the backward kernels of all three grids hand their d enc over
    g_enc = (torch.randn((N * S, net.num_levels, 2), device=dev) * 1e-3).permute(1, 0, 2).contiguous()
    # the proposal grids' scatter computes d position itself; the main field's does not (k_field_dpos does, from the saved d enc / d offset)
    with_dpos = net.num_levels == 5
    for _ in range(R):
        # (as the shared-mode step runs it: the gradients are zero when the scatter starts, the fold stores -- TnGrid.table_grad_is_zero)
        ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions, L.e_bins, g_enc,
                         d_o if with_dpos else None, d_d if with_dpos else None, grad_is_zero=True)
    torch.cuda.synchronize()
gd = torch.rand_like(lv[2].density) * 1e-2; gc = torch.rand_like(b.rgb_samples)
ph = ops._lib
for _ in range(R):
    ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv[2].e_bins, gd, gc, d_o, d_d, ph.TN_BWD_MLP | ph.TN_BWD_JOIN)  # + k_field_dpos
    torch.cuda.synchronize()
eng.arena.zero_grad()
a = eng.arena
for _ in range(R):
    ops.adam_step_ranges(a.params, a.grads, a.exp_avg, a.exp_avg_sq, [a.group_range[g] + (1, 1e-2) for g in a.optimised_groups], eps=1e-15)
torch.cuda.synchronize()
