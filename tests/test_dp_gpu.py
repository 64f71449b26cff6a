"""Data-parallel step on the GPU: the backward in phases (tn_field_bwd_phase) with the gradient exchange overlapped
(parallel.OverlappedGradReducer) must produce the gradients and the parameter update of the plain step.  The GPU box has one GPU, so the
process group has one rank: every collective still goes through RCCL (backend "nccl"), issued from the streams the real schedule uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from nerfstudio_thermal_amd import _lib, ops
from nerfstudio_thermal_amd.parallel import GradAllReducer, OverlappedGradReducer
from test_model_gpu import build, dev_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def rccl_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def run_steps(golden_dir, mode, hook, steps=3):
    ocfg, cfg, arena, eng = build(mode)
    gi, o, d, cam = dev_inputs(golden_dir)
    img, is_th = gi["image"].to(DEV), gi["is_thermal"].to(DEV)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]
    grads, losses = None, None
    for step in range(steps):
        losses = eng.train_step(o, d, cam, img, is_th, step, jitters=jit, jitters_thermal=jit_t, grad_hook=hook)
        if step == 0:
            grads = arena.grads.clone()
    torch.cuda.synchronize()
    return arena, grads, {k: float(v) for k, v in losses.items()}


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_overlapped_exchange_matches_plain_step(golden_dir, rccl_group, mode):
    a0, g0, l0 = run_steps(golden_dir, mode, None)
    a1, g1, l1 = run_steps(golden_dir, mode, OverlappedGradReducer(1, level_chunks=4))
    a2, g2, l2 = run_steps(golden_dir, mode, GradAllReducer(1))
    scale = float(g0.abs().max())
    # float atomics make two runs of the SAME schedule differ in the last bits; the phased schedule must be inside that noise
    noise = float((g0 - g2).abs().max())
    assert float((g0 - g1).abs().max()) <= max(4.0 * noise, 1e-6 * scale), (float((g0 - g1).abs().max()), noise, scale)
    assert torch.equal(g0 == 0, g1 == 0)  # identical sparsity (Adam eps=1e-15 makes exact zeros matter)
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-4 * abs(l0[k]) + 1e-9, (k, l0[k], l1[k])
    # three Adam steps later the parameters agree to the step-to-step noise level as well.  Adam (eps = 1e-15) turns last-bit gradient
    # differences of near-zero entries into lr-sized parameter differences, so single entries are noisy: compare in the L2 sense.
    # Two runs of the SAME schedule differ by 0.05-0.5 % of the distance travelled (bimodal: one sample crossing a cell boundary after
    # the first update changes the rest), so the phased schedule is held to 2 % of it.
    dp = float((a0.params - a1.params).double().norm())
    dn = float((a0.params - a2.params).double().norm())
    moved = float((a0.params - build(mode)[2].params).double().norm())
    assert dp <= 0.02 * moved, (dp, dn, moved)


def test_field_bwd_phases_equal_whole(golden_dir):
    """tn_field_bwd_phase(MLP) + SCATTER over level ranges + JOIN == tn_field_bwd, entry point by entry point."""
    ocfg, cfg, arena, eng = build("shared")
    gi, o, d, cam = dev_inputs(golden_dir)
    out, br = eng.get_outputs(o, d, cam, True, [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]])
    b = br[""]
    lv = b.levels[2]
    gd = torch.rand_like(lv.density)
    gc = torch.rand_like(b.rgb_samples)
    res = []
    # whole / SCATTER per level range (bin + fold each) / one SCATTER_BIN over all levels + SCATTER_FOLD per level range
    for ranges, two_step in ((None, False), ([(0, 16)], False), ([(0, 5), (5, 6), (6, 16)], False), ([(0, 6), (6, 10), (10, 13), (13, 15), (15, 16)], True)):
        arena.zero_grad()
        d_o, d_d = torch.zeros_like(o), torch.zeros_like(d)
        if ranges is None:
            ops.field_bwd(eng.field, b.origins, b.directions, cam, lv.e_bins, gd, gc, d_o, d_d)
        else:
            ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv.e_bins, gd, gc, d_o, d_d, _lib.TN_BWD_MLP)
            if two_step:
                ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv.e_bins, gd, gc, d_o, d_d, _lib.TN_BWD_SCATTER_BIN)
            for lb, le in ranges:
                ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv.e_bins, gd, gc, d_o, d_d,
                                    _lib.TN_BWD_SCATTER_FOLD if two_step else _lib.TN_BWD_SCATTER, lb, le)
            ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv.e_bins, gd, gc, d_o, d_d, _lib.TN_BWD_JOIN)
        torch.cuda.synchronize()
        res.append((arena.grads.clone(), d_o.clone(), d_d.clone()))
    for k in (1, 2, 3):
        for a, bb in zip(res[0], res[k]):
            assert float((a - bb).abs().max()) <= 2e-6 * float(a.abs().max()), k
    with pytest.raises(RuntimeError):
        ops.field_bwd_phase(eng.field, b.origins, b.directions, cam, lv.e_bins, gd, gc, None, None, _lib.TN_BWD_SCATTER, 3, 17)


def test_dense_exchange_of_coarse_levels_full_size(rccl_group):
    """BASELINE config 1 sizes: the coarse table levels (0-4 at 4096 rays) go through tn_field_bwd_scatter_dense -> all-reduce of the per-cell
    sums -> tn_field_dense_fold instead of the 20 MB table slice.  Gradients (zero pattern included) must equal the plain backward's."""
    import bench
    from nerfstudio_thermal_amd import synth

    dev = torch.device(DEV, 0)
    cam_t, idx, img, is_th = bench.make_batch(dev, 4096, 42)
    o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
    cam = idx[:, 0].contiguous()
    jit = [torch.from_numpy(j).to(dev).reshape(-1).contiguous() for j in synth.synth_jitters(4096, seed=5)]
    grads = []
    for hook in (None, OverlappedGradReducer(1), OverlappedGradReducer(1, dense_exchange=True)):  # plain level ranges (default) / dense coarse levels
        cfg, arena, eng = bench.build_engine(dev)
        assert ops.field_dense_count(eng.field, 4096 * 48, 0, 5) == 4913 + 12167 + 29791 + 79507 + 205379
        assert ops.field_dense_count(eng.field, 4096 * 48, 0, 6) == 0  # level 5 (res 80) has more cells than the table has slots
        arena.zero_grad()
        out, branches = eng.get_outputs(o, d, cam, True, jit)
        if hook is None:
            eng.loss_and_backward(out, branches, cam, img, is_th)
        else:
            hook.begin(arena)
            eng.loss_and_backward(out, branches, cam, img, is_th, dp=hook)
            covered = list(hook.finish_iter())
            lo, hi = arena.live_range
            assert sorted(covered)[0][0] == lo and sorted(covered)[-1][1] == hi
        torch.cuda.synchronize()
        grads.append(arena.grads.clone())
    g0 = grads[0]
    scale = float(g0.abs().max())
    for g1 in grads[1:]:
        assert float((g0 - g1).abs().max()) <= 2e-6 * scale, float((g0 - g1).abs().max()) / scale
        assert int(((g0 == 0) != (g1 == 0)).sum()) <= 4  # exact cancellations may leave a residue in another summation order


def test_sharded_reducer_and_bf16_transport_on_one_rank(golden_dir, rccl_group):
    """The two data-parallel variants on a 1-rank RCCL group (what a GPU box with one GPU can run): the sharded-optimiser schedule degenerates to
    the per-range Adam schedule (world 1: every rank owns everything) and must match the plain step; bf16 transport rounds the big slices'
    gradients to 8 mantissa bits -- the first step's gradients agree to 2^-7 relative, exact zeros stay exact zeros."""
    from nerfstudio_thermal_amd.parallel import ShardedGradReducer

    # (reference: the per-range Adam schedule, which like the sharded one does not consume the gradients -- the fused step's single Adam launch
    # zeroes them behind its read, so its gradient buffer is all zero afterwards)
    ref = OverlappedGradReducer(1, level_chunks=2)
    ref.adam_per_range = True
    a0, g0, l0 = run_steps(golden_dir, "shared", ref)
    a1, g1, l1 = run_steps(golden_dir, "shared", ShardedGradReducer(1, 0, level_chunks=2))
    scale = float(g0.abs().max())
    assert scale > 0
    noise = float((g0 - run_steps(golden_dir, "shared", ref, steps=1)[1]).abs().max())
    assert float((g0 - g1).abs().max()) <= max(4.0 * noise, 1e-6 * scale)
    moved = float((a0.params - build("shared")[2].params).double().norm())
    assert float((a0.params - a1.params).double().norm()) <= 0.02 * moved
    hook = OverlappedGradReducer(1, level_chunks=2, transport_dtype=torch.bfloat16)
    hook.adam_per_range = True
    a2, g2, l2 = run_steps(golden_dir, "shared", hook, steps=1)
    assert torch.equal(g0 == 0, g2 == 0)
    # (the tiny test tables are below the 1 M-gradient bar of the conversion: nothing was rounded, the schedule itself must be unchanged)
    assert float((g0 - g2).abs().max()) <= max(4.0 * noise, 1e-6 * scale)


def test_allreduce_grads_through_the_c_abi_on_a_one_rank_communicator():
    """tn_comm_unique_id / tn_comm_create / tn_allreduce_grads / tn_comm_destroy: an RCCL communicator and the gradient exchange without
    torch.distributed -- the reference's DDP mean all-reduce (pipelines/base_pipeline.py:281-283) for a host that binds only the C ABI.  One rank
    is what a one-GPU box offers (RCCL refuses two ranks on one device): the mean over one rank and the sum are the identity, on the caller's
    stream, in place, on a slice of a gradient arena."""
    from nerfstudio_thermal_amd import ops

    comm = ops.RcclComm(ops.RcclComm.unique_id(), 1, 0)
    try:
        g = torch.randn(3_000_001, device="cuda")
        want = g.clone()
        comm.allreduce_grads(g[1:2_000_001])                 # a slice (a level range of the table), mean
        comm.allreduce_grads(g[2_000_001:], average=False)   # another one, sum
        torch.cuda.synchronize()
        assert torch.equal(g, want)
        with pytest.raises(ValueError):
            comm.allreduce_grads(g.cpu())
    finally:
        comm.destroy()
