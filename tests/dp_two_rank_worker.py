"""One rank of tests/test_dp_two_ranks_gpu.py: the data-parallel train step (engine.train_step + parallel.OverlappedGradReducer, world size 2) on
REAL kernels.  A GPU box has one GPU, so both ranks drive cuda:0 and the collectives go through gloo (RCCL refuses two ranks on one device);
everything around the collective -- the phased backward, the level ranges, the second communicator for the proposal networks, the order of the
exchanges, the sum / scale, Adam on the exchanged gradients -- is the code a multi-GPU run executes.

    python tests/dp_two_rank_worker.py <rank> <world> <port> <golden_dir> <out.json>
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main_ddp(rank, world, port, golden_dir, out_path):
    """The reference's own multi-GPU path on the drop-in classes: DistributedDataParallel(model, find_unused_parameters=True)
    (pipelines/base_pipeline.py:281-283) around the Trainer's iteration (engine/trainer.py:455-499), two ranks with different batches."""
    from torch.nn.parallel import DistributedDataParallel as DDP

    from nerfstudio_thermal_amd.rays import RayBundle
    from test_trainer_sequence_gpu import _reference_train_iteration, _setup

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mode = os.environ.get("TN_TEST_DENSITY_MODE", "shared")
    model, opt, rb, batch, _ = _setup(golden_dir, mode)
    init = model.arena.params.detach().clone()
    sh = 16 * rank  # this rank's rays: the golden batch rolled by whole 2x2 patches, ground truth rolled with them
    rb = RayBundle(origins=torch.roll(rb.origins, sh, 0).contiguous(), directions=torch.roll(rb.directions, sh, 0).contiguous(),
                   pixel_area=torch.roll(rb.pixel_area, sh, 0).contiguous(), camera_indices=torch.roll(rb.camera_indices, sh, 0).contiguous())
    batch = {k: torch.roll(v, sh, 0).contiguous() for k, v in batch.items()}
    ddp = DDP(model, device_ids=[torch.cuda.current_device()], find_unused_parameters=True)  # (broadcasts rank 0's parameters and buffers)
    scaler = torch.amp.GradScaler("cuda")
    seen_idle = False
    for step in range(13):  # (the first iteration without a proposal update is the 11th)
        torch.manual_seed(100 + step)  # the ranks draw the same jitter: their gradients differ through the batches alone
        model.engine.__dict__.pop("_rand", None)
        losses, _ = _reference_train_iteration(model, opt, scaler, rb, batch, step, True, call=ddp)
        seen_idle = seen_idle or not model.engine.last_updated
    torch.cuda.synchronize()
    mine = model.arena.params.detach().cpu()
    theirs = mine.clone()
    dist.broadcast(theirs, src=0)
    res = {"rank": rank, "params_equal_rank0": bool(torch.equal(mine, theirs)), "params_finite": bool(torch.isfinite(mine).all()),
           "moved": float((model.arena.params - init).double().norm()), "seen_idle": bool(seen_idle), "scale": float(scaler.get_scale()),
           "losses": {k: float(v) for k, v in losses.items()}}
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def main_fused_trainer(rank, world, port, golden_dir, out_path):
    """trainer.FusedTrainerMixin with two ranks: the Trainer's iteration on the fused step, the gradient exchange issued by the mixin
    (parallel.OverlappedGradReducer / GradAllReducer over the default process group, picked by parallel.InRunScheduleGuard from the run's first
    twelve iterations) instead of DistributedDataParallel's autograd hooks."""
    from nerfstudio_thermal_amd.parallel import broadcast_params
    from nerfstudio_thermal_amd.rays import RayBundle
    from nerfstudio_thermal_amd.trainer import FusedTrainerMixin
    from test_fused_trainer_gpu import RefTrainer
    from test_trainer_sequence_gpu import _setup

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mode = os.environ.get("TN_TEST_DENSITY_MODE", "shared")
    model, opt, rb, batch, _ = _setup(golden_dir, mode)
    init = model.arena.params.detach().clone()
    broadcast_params(model.arena)  # (what the DistributedDataParallel wrap of the pipeline does at construction)
    sh = 16 * rank
    rb = RayBundle(origins=torch.roll(rb.origins, sh, 0).contiguous(), directions=torch.roll(rb.directions, sh, 0).contiguous(),
                   pixel_area=torch.roll(rb.pixel_area, sh, 0).contiguous(), camera_indices=torch.roll(rb.camera_indices, sh, 0).contiguous())
    batch = {k: torch.roll(v, sh, 0).contiguous() for k, v in batch.items()}
    trainer = type("HipTrainer", (FusedTrainerMixin, RefTrainer), {})(model, opt, rb, batch)
    loss, loss_dict, metrics = trainer.loop(range(13))
    torch.cuda.synchronize()
    mine = model.arena.params.detach().cpu()
    theirs = mine.clone()
    dist.broadcast(theirs, src=0)
    res = {"rank": rank, "params_equal_rank0": bool(torch.equal(mine, theirs)), "params_finite": bool(torch.isfinite(mine).all()),
           "moved": float((model.arena.params - init).double().norm()), "seen_idle": True, "scale": 65536.0,
           "reference_iterations": trainer.ref_iterations, "exchanges": type(trainer.__dict__["_tn_dp_guard"].hooks["overlapped"]).__name__,
           "guard": {"schedule": trainer.__dict__["_tn_dp_guard"].decision["schedule"], "legs": [len(v) for v in trainer.__dict__["_tn_dp_guard"].times.values()]},
           "losses": {k: float(v) for k, v in loss_dict.items()}}
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world, port, golden_dir, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    if os.environ.get("TN_TEST_REDUCER", "overlapped") == "ddp":
        return main_ddp(rank, world, port, golden_dir, out_path)
    if os.environ.get("TN_TEST_REDUCER", "overlapped") == "fused_trainer":
        return main_fused_trainer(rank, world, port, golden_dir, out_path)
    from test_model_gpu import build, dev_inputs

    from nerfstudio_thermal_amd.parallel import OverlappedGradReducer, broadcast_params

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mode = os.environ.get("TN_TEST_DENSITY_MODE", "shared")
    ocfg, cfg, arena, eng = build(mode)
    gi, o, d, cam = dev_inputs(golden_dir)
    img, is_th = gi["image"].to("cuda"), gi["is_thermal"].to("cuda")
    jit = [j.to("cuda").reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to("cuda").reshape(-1).contiguous() for j in gi["jitters_thermal"]]

    def batch(r):
        # rank r's rays: the golden batch rolled by 16 r rays (whole 2x2 patches), ground truth rolled with them
        return tuple(torch.roll(t, 16 * r, 0).contiguous() for t in (o, d, cam, img, is_th))

    broadcast_params(arena)
    # what the exchange must deliver: the mean of the ranks' gradients -- every rank's batch through the PLAIN backward here, same parameters, same jitters
    ref = []
    eng.set_anneal_for_step(0)  # (what train_step does first: the proposal weights' anneal exponent of iteration 0)
    for r in range(world):
        bo, bd, bc, bi, bt = batch(r)
        arena.zero_grad()
        outp, br = eng.get_outputs(bo, bd, bc, True, jit, jit_t)
        eng.loss_and_backward(outp, br, bc, bi, bt)
        torch.cuda.synchronize()
        ref.append(arena.grads.clone())
    arena.zero_grad()
    expected = sum(ref) / world
    noise_scale = float(expected.abs().max())

    class Capture(OverlappedGradReducer):
        captured = None

        def finish(self, skip=None):
            super().finish(skip)
            if self.captured is None:
                torch.cuda.synchronize()
                self.captured = self._arena.grads.clone()

    hook = Capture(world, level_chunks=(6, 6, 4))
    bo, bd, bc, bi, bt = batch(rank)
    losses = None
    for step in range(3):
        losses = eng.train_step(bo, bd, bc, bi, bt, step, jitters=jit, jitters_thermal=jit_t, grad_hook=hook)
    torch.cuda.synchronize()
    got = hook.captured
    # parameters after three steps: every rank must hold rank 0's, bit for bit
    mine = arena.params.detach().cpu()
    theirs = mine.clone()
    dist.broadcast(theirs, src=0)
    per_group = {}
    for g in arena.optimised_groups:
        lo, hi = arena.group_range[g]
        sl = slice(lo, hi)
        per_group[g] = {"mean": float((got[sl] - expected[sl]).abs().max()), "sum": float((got[sl] - world * expected[sl]).abs().max()),
                        "own": float((got[sl] - ref[rank][sl]).abs().max()), "scale": float(expected[sl].abs().max())}
    res = {
        "rank": rank,
        "per_group": per_group,
        "grad_err": float((got - expected).abs().max()),
        "grad_scale": noise_scale,
        "own_vs_mean": float((ref[rank] - expected).abs().max()),  # the ranks' gradients DO differ: the exchange is not vacuous
        "zero_mismatch": int(((got == 0) != (expected == 0)).sum()),
        "params_equal_rank0": bool(torch.equal(mine, theirs)),
        "params_finite": bool(torch.isfinite(mine).all()),
        "losses": {k: float(v) for k, v in losses.items()},
    }
    if os.environ.get("TN_TEST_REDUCER", "overlapped") == "sharded":
        # the sharded-optimiser schedule (reduce-scatter -> Adam on the owned 1/world piece -> all-gather of the parameters; gloo: all-reduce +
        # all-gather) from the same start: bit-identical parameters on every rank, and the all-reduce schedule's parameters up to the noise of
        # two runs (Adam with eps 1e-15 turns last-bit gradient differences into lr-sized steps on near-zero entries: compared in the L2 sense)
        from nerfstudio_thermal_amd.parallel import ShardedGradReducer

        ocfg2, cfg2, arena2, eng2 = build(mode)
        init = arena2.params.detach().clone()
        broadcast_params(arena2)
        class Counting(ShardedGradReducer):
            slices = 0

            def gather_params(self):
                self.slices += len(self._gather)  # slices that were reduce-scattered, updated by their owner and gathered back
                super().gather_params()

        sh = Counting(world, rank, min_shard=1 << 10, level_chunks=(6, 6, 4))
        for step in range(3):
            eng2.train_step(bo, bd, bc, bi, bt, step, jitters=jit, jitters_thermal=jit_t, grad_hook=sh)
        torch.cuda.synchronize()
        mine2 = arena2.params.detach().cpu()
        theirs2 = mine2.clone()
        dist.broadcast(theirs2, src=0)
        res["sharded"] = {"params_equal_rank0": bool(torch.equal(mine2, theirs2)), "params_finite": bool(torch.isfinite(mine2).all()),
                          "dist_to_allreduce": float((arena2.params - arena.params).double().norm()),
                          "moved": float((arena.params - init).double().norm()),
                          "sharded_slices": sh.slices}
        # GradScaler semantics on the sharded schedule (ShardedGradReducer.reduce_flags): iteration 3 sees an inf in ONE rank's ground truth only.
        # The non-finite gradient reaches the owners of the affected pieces -- and every rank must skip the affected groups all the same: parameters
        # and moments untouched there, bit-identical across the ranks, the scale halved and the skip counted on both; iteration 4 steps again.
        from nerfstudio_thermal_amd.optim import DeviceGradScaler

        sc = DeviceGradScaler("cuda", num_groups=len(arena2.optimised_groups))
        eng2.train_step(bo, bd, bc, bi, bt, 3, jitters=jit, jitters_thermal=jit_t, grad_hook=sh, grad_scaler=sc)  # a clean scaled iteration first
        torch.cuda.synchronize()
        before = (arena2.params.clone(), arena2.exp_avg.clone(), arena2.exp_avg_sq.clone())
        bad = bi.clone()
        if rank == 1:
            bad[:8] = float("inf")
        eng2.train_step(bo, bd, bc, bad, bt, 4, jitters=jit, jitters_thermal=jit_t, grad_hook=sh, grad_scaler=sc)
        torch.cuda.synchronize()
        lo, hi = arena2.group_range["fields"]
        untouched = all(bool(torch.equal(x[lo:hi], y[lo:hi])) for x, y in zip(before, (arena2.params, arena2.exp_avg, arena2.exp_avg_sq)))
        skipped = [sc.num_skipped(i) for i in range(len(arena2.optimised_groups))]
        scale_after = sc.get_scale()
        eng2.train_step(bo, bd, bc, bi, bt, 5, jitters=jit, jitters_thermal=jit_t, grad_hook=sh, grad_scaler=sc)
        torch.cuda.synchronize()
        mine3 = arena2.params.detach().cpu()
        theirs3 = mine3.clone()
        dist.broadcast(theirs3, src=0)
        res["sharded_scaler"] = {"fields_untouched_on_inf": untouched, "skipped": skipped, "scale_after_inf": scale_after,
                                 "params_equal_rank0": bool(torch.equal(mine3, theirs3)), "params_finite": bool(torch.isfinite(mine3).all()),
                                 "moved_after": float((arena2.params - before[0]).double().norm())}
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
