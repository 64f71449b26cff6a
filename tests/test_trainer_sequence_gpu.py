"""The reference Trainer's ACTUAL step on the drop-in objects (engine/trainer.py:463-495, mixed_precision=True in thermal-nerfacto's method config,
configs/method_configs.py:260): torch.autocast around the forward, grad_scaler.scale(loss).backward(), Optimizers.optimizer_scaler_step_some
(unscale + max_norm clip + GradScaler.step), grad_scaler.update(), schedulers stepped only when the scale did not drop -- on HipFusedAdam
(skip decided on the device) and on torch.optim.Adam; the same semantics on the fused step (optim.DeviceGradScaler); and the reference's
multi-GPU wrap DistributedDataParallel(model, find_unused_parameters=True) (pipelines/base_pipeline.py:281-283) on a 1-rank RCCL group."""
import functools
import os

import numpy as np
import pytest
import torch

from helpers import make_params
from test_model_api_gpu import DEV, build_model, bundle

pytestmark = pytest.mark.gpu


def _reference_train_iteration(model, optimizers, grad_scaler, rb, batch, step, mixed_precision=True, jitters=(None, None), call=None):
    """engine/trainer.py:455-499, statement for statement (gradient_accumulation_steps = 1 for every group)."""
    from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc

    cbs = model.get_training_callbacks()
    for cb in cbs:
        cb.run_callback_at_location(step, Loc.BEFORE_TRAIN_ITERATION)
    optimizers.zero_grad_some(list(optimizers.optimizers.keys()))
    with torch.autocast(device_type="cuda", enabled=mixed_precision):
        if call is not None:
            out = call(rb[...])  # e.g. the DistributedDataParallel wrapper
        else:
            out = model.get_outputs(model.collider(rb[...]), *jitters)
        metrics = model.get_metrics_dict(out, batch)
        loss_dict = model.get_loss_dict(out, batch, metrics)
        loss = functools.reduce(torch.add, loss_dict.values())
    grad_scaler.scale(loss).backward()
    optimizers.optimizer_scaler_step_some(grad_scaler, list(optimizers.optimizers.keys()))
    scale = grad_scaler.get_scale()
    grad_scaler.update()
    stepped_sched = False
    if scale <= grad_scaler.get_scale():  # "If the gradient scaler is decreased, no optimization step is performed so we should not step the scheduler"
        optimizers.scheduler_step_all(step)
        stepped_sched = True
    for cb in cbs:
        cb.run_callback_at_location(step, Loc.AFTER_TRAIN_ITERATION)
    return loss_dict, stepped_sched


def _setup(golden_dir, mode, optimizer_cls=None, max_norm=None):
    from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers

    from helpers import golden_inputs  # noqa: F401

    gi, rb = bundle(golden_dir)
    batch = {"image": gi["image"].to(DEV).clone(), "is_thermal": gi["is_thermal"].to(DEV)}
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]
    ocfg, cfg, model = build_model(mode)
    model.arena.load(make_params(ocfg))
    model.train()
    opt = Optimizers(model.get_param_groups(), optimizer_cls=optimizer_cls or HipFusedAdam, max_norm=max_norm)
    return model, opt, rb, batch, (jit, jit_t)


def _snapshot(model):
    a = model.arena
    return a.params.clone(), a.exp_avg.clone(), a.exp_avg_sq.clone()


def assert_same_training_state(A, B, what="", moments=True, frac=1e-2):
    """Two runs of the same iterations.  The gradients of two runs differ in their last bits (float atomics, record order inside a bucket), and
    Adam with eps = 1e-15 turns a sign flip of a noise-level gradient entry into a full lr step: equality is asserted on the moments
    (when both runs keep them in the arena) and on all but a small fraction of the parameters."""
    (pA, mA, vA), (pB, mB, vB) = A, B
    pairs = [("params", pA, pB, 1e-5)]
    if moments:
        pairs += [("exp_avg", mA, mB, 1e-3 * float(mB.abs().max()) + 1e-30), ("exp_avg_sq", vA, vB, 1e-3 * float(vB.abs().max()) + 1e-30)]
    for name, x, y, atol in pairs:
        off = ((x - y).abs() > atol).float().mean()
        assert float(off) <= frac, (what, name, float(off))


def _sync(dst_model, dst_opt, src_model, src_opt):
    """dst := src (parameters, optimiser state through state_dict / load_state_dict, schedulers, sampler counters): two runs are compared one
    iteration at a time from identical states, because over many iterations the sign flips above make any two runs drift apart."""
    import copy

    dst_model.arena.params.copy_(src_model.arena.params)
    for g in src_opt.optimizers:
        dst_opt.optimizers[g].load_state_dict(copy.deepcopy(src_opt.optimizers[g].state_dict()))
        dst_opt.schedulers[g].load_state_dict(src_opt.schedulers[g].state_dict())
    for k in ("steps_since_update", "sampler_step", "anneal"):
        setattr(dst_model.engine, k, getattr(src_model.engine, k))


def _group_slices(model):
    a = model.arena
    return {g: slice(*a.group_range[g]) for g in a.optimised_groups}


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_scaler_sequence_hip_vs_torch_adam_and_skip_on_inf(golden_dir, mode):
    """A: HipFusedAdam under the trainer's autocast + GradScaler sequence (skip decided on the device); B: torch.optim.Adam under the same
    sequence (GradScaler's own found_inf.item() path: what the reference runs); C: HipFusedAdam without autocast / scaler."""
    from nerfstudio_thermal_amd.optim import HipFusedAdam

    mA, oA, rb, batch, jit = _setup(golden_dir, mode, HipFusedAdam)
    mB, oB, _, _, _ = _setup(golden_dir, mode, torch.optim.Adam)
    mC, oC, _, _, _ = _setup(golden_dir, mode, HipFusedAdam)
    sA, sB = torch.amp.GradScaler("cuda", enabled=True), torch.amp.GradScaler("cuda", enabled=True)
    sC = torch.amp.GradScaler("cuda", enabled=False)
    for step in range(3):
        _sync(mB, oB, mA, oA)
        _sync(mC, oC, mA, oA)
        lA, _ = _reference_train_iteration(mA, oA, sA, rb, batch, step, True, jit)
        lB, _ = _reference_train_iteration(mB, oB, sB, rb, batch, step, True, jit)
        lC, _ = _reference_train_iteration(mC, oC, sC, rb, batch, step, False, jit)
        for k in lC:
            for other in (lA, lB):
                assert abs(float(other[k].detach()) - float(lC[k].detach())) <= (1e-6 if step == 0 else 2e-4) * abs(float(lC[k].detach())) + 1e-12, (step, k)
    # scaling by 2^16 and unscaling is exact in fp32; the AMP Adam evaluates its bias corrections on the device (last-ulp differences at most)
    assert_same_training_state(_snapshot(mA), _snapshot(mC), "3 iterations with / without autocast + GradScaler")
    assert_same_training_state(_snapshot(mA), _snapshot(mB), "HipFusedAdam vs torch.optim.Adam under the scaler", moments=False)
    assert sA.get_scale() == 65536.0
    # ---- a forced-inf iteration.  The inf enters through the pixel loss: it reaches the field and the pose, NOT the proposal networks (their
    # gradient comes from the interlevel loss).  GradScaler decides per optimiser: `fields*` and `camera_opt*` must not move (parameters and both
    # moments bit-identical), the proposal networks take their step; the scale halves and the trainer does not step the schedulers.
    lr_before = {k: s.get_last_lr()[0] for k, s in oA.schedulers.items()}
    bad = dict(batch)
    bad["image"] = batch["image"].clone()
    bad["image"][:, :] = float("inf")
    _sync(mB, oB, mA, oA)
    for m_, o_, s_ in ((mA, oA, sA), (mB, oB, sB)):
        before = _snapshot(m_)
        _, stepped = _reference_train_iteration(m_, o_, s_, rb, bad, 3, True, jit)
        after = _snapshot(m_)
        for g, sl in _group_slices(m_).items():
            same = [bool(torch.equal(b[sl], a[sl])) for b, a in zip(before, after)]
            if g.startswith("proposal_networks"):
                assert not same[0], g  # stepped
            else:
                assert all(same), (g, same)
        assert not stepped and s_.get_scale() == 32768.0
    assert {k: s.get_last_lr()[0] for k, s in oA.schedulers.items()} == lr_before
    assert {n: o.num_skipped() for n, o in oA.optimizers.items()} == {n: (0 if n.startswith("proposal_networks") else 1) for n in oA.optimizers}
    # ---- the next clean iteration: the device-side step counts (bias correction) of A must be those of torch's step tensors in B
    mB.arena.params.copy_(mA.arena.params)  # (only the parameters: B keeps ITS optimiser state, whose step tensors are what is being compared)
    _reference_train_iteration(mA, oA, sA, rb, batch, 4, True, jit)
    _reference_train_iteration(mB, oB, sB, rb, batch, 4, True, jit)
    assert_same_training_state(_snapshot(mA), _snapshot(mB), "the iteration after a skipped one", moments=False)
    for name in oA.optimizers:
        kA = sorted({int(v["step"]) for v in oA.optimizers[name].state_dict()["state"].values()})
        kB = sorted({int(v["step"]) for v in oB.optimizers[name].state_dict()["state"].values()})
        assert kA == kB == ([5] if name.startswith("proposal_networks") else [4]), (name, kA, kB)


def test_two_scaler_steps_in_one_iteration_keep_each_others_found_inf(golden_dir):
    """optimizer_scaler_step_some called TWICE before grad_scaler.update() (two disjoint group lists): the flags the first call raised must
    still be there when update() collects them -- the per-optimiser found_inf views alias one buffer, and the second call clears only the
    entries of the optimisers it steps."""
    import functools

    from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc
    from nerfstudio_thermal_amd.optim import HipFusedAdam

    m, o, rb, batch, jit = _setup(golden_dir, "shared", HipFusedAdam)
    s = torch.amp.GradScaler("cuda")
    _reference_train_iteration(m, o, s, rb, batch, 0, True, jit)  # a clean iteration first (optimiser state exists)
    bad = dict(batch)
    bad["image"] = torch.full_like(batch["image"], float("inf"))  # reaches fields + camera_opt, not the proposal networks
    for cb in m.get_training_callbacks():
        cb.run_callback_at_location(1, Loc.BEFORE_TRAIN_ITERATION)
    o.zero_grad_some(list(o.optimizers.keys()))
    with torch.autocast(device_type="cuda"):
        out = m.get_outputs(m.collider(rb[...]), *jit)
        loss = functools.reduce(torch.add, m.get_loss_dict(out, bad, m.get_metrics_dict(out, bad)).values())
    s.scale(loss).backward()
    first = [g for g in o.optimizers if not g.startswith("proposal_networks")]
    second = [g for g in o.optimizers if g.startswith("proposal_networks")]
    before = _snapshot(m)
    o.optimizer_scaler_step_some(s, first)   # raises found_inf for fields / camera_opt
    o.optimizer_scaler_step_some(s, second)  # finite gradients: must not wipe the flags above
    s.update()
    assert s.get_scale() == 32768.0, "the non-finite gradients of the first call were forgotten"
    after = _snapshot(m)
    for g, sl in _group_slices(m).items():
        same = all(bool(torch.equal(b[sl], a[sl])) for b, a in zip(before, after))
        assert same != g.startswith("proposal_networks"), g


def test_max_norm_clipping_matches_torch(golden_dir):
    """engine/optimizers.py:160-173 with OptimizerConfig.max_norm set: unscale_, clip_grad_norm_, then the step; HipFusedAdam against torch.optim.Adam."""
    from nerfstudio_thermal_amd.optim import HipFusedAdam

    mn = {"fields": 1e-4, "proposal_networks": 1e-3}
    mA, oA, rb, batch, jit = _setup(golden_dir, "shared", HipFusedAdam, mn)
    mB, oB, _, _, _ = _setup(golden_dir, "shared", torch.optim.Adam, mn)
    sA, sB = torch.amp.GradScaler("cuda"), torch.amp.GradScaler("cuda")
    for step in range(2):
        _reference_train_iteration(mA, oA, sA, rb, batch, step, True, jit)
        _reference_train_iteration(mB, oB, sB, rb, batch, step, True, jit)
        for name in ("fields", "proposal_networks"):
            tot = torch.sqrt(sum((p.grad.detach() ** 2).sum() for p in oA.parameters[name] if p.grad is not None))
            assert float(tot) <= mn[name] * (1 + 1e-4), (name, float(tot))  # gradients were unscaled and clipped in place
    assert_same_training_state(_snapshot(mA), _snapshot(mB), "max_norm: HipFusedAdam vs torch.optim.Adam", moments=False)


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_fused_step_grad_scaler_semantics(golden_dir, mode):
    """RenderEngine.train_step with optim.DeviceGradScaler against the drop-in path under torch.amp.GradScaler (HipFusedAdam): clean steps, a
    forced-inf step (per-group skip, scale backoff, LR schedule not advanced: all on the device), and the step after it."""
    from nerfstudio_thermal_amd.optim import DeviceGradScaler, HipFusedAdam

    mA, _, rb, batch, jit = _setup(golden_dir, mode)
    rays = (rb.origins.contiguous(), rb.directions.contiguous(), rb.camera_indices.reshape(-1).contiguous())
    eA = mA.engine
    mB, oB, _, _, _ = _setup(golden_dir, mode, HipFusedAdam)
    mC, _, _, _, _ = _setup(golden_dir, mode)
    sA, sB = DeviceGradScaler(DEV), torch.amp.GradScaler("cuda")
    bad = batch["image"].clone()
    bad[:, :] = float("inf")
    G = mA.arena.optimised_groups
    for step in range(4):
        img = bad if step == 2 else batch["image"]
        before = _snapshot(mA)
        eA.train_step(*rays, img, batch["is_thermal"], step, jit[0], jit[1], grad_scaler=sA)
        _reference_train_iteration(mB, oB, sB, rb, {"image": img, "is_thermal": batch["is_thermal"]}, step, True, jit)
        if step < 2:
            mC.engine.train_step(*rays, img, batch["is_thermal"], step, jit[0], jit[1])
        if step == 1:
            assert_same_training_state(_snapshot(mA), _snapshot(mC), "fused step with / without the device grad scaler")
        if step == 2:
            after = _snapshot(mA)
            for g, sl in _group_slices(mA).items():
                same = [bool(torch.equal(b[sl], a[sl])) for b, a in zip(before, after)]
                assert (not same[0]) if g.startswith("proposal_networks") else all(same), (g, same)
            assert sA.get_scale() == 32768.0 == sB.get_scale() and sA.schedule_lag() == 1
            assert [sA.num_skipped(i) for i in range(len(G))] == [0 if g.startswith("proposal_networks") else 1 for g in G]
        # (four iterations WITHOUT re-synchronising the two runs: the sign flips accumulate, a few percent of the entries drift)
        assert_same_training_state(_snapshot(mA), _snapshot(mB), f"fused step vs drop-in path under GradScaler, iteration {step}", frac=5e-2)
    # the drop-in trainer did not step its schedulers in iteration 2; the fused step's device-side schedule lags by one as well:
    # lr used by the fused step in iteration 3 = schedule(3 - 1) = what the drop-in schedulers hold after three scheduler steps
    from nerfstudio_thermal_amd.engine import OPTIMIZERS, exp_decay_lr

    assert abs(oB.schedulers["fields"].get_last_lr()[0] - exp_decay_lr(3, *OPTIMIZERS["fields"])) <= 1e-12


def test_growth_interval_and_partial_group_skip():
    """GradScaler.update()'s growth and the per-OPTIMISER skip decision (torch/amp/grad_scaler.py) on the device: an inf in one group's gradients
    skips that group only, the scale still backs off."""
    from nerfstudio_thermal_amd import ops
    from nerfstudio_thermal_amd.optim import DeviceGradScaler

    s = DeviceGradScaler(DEV, num_groups=2, init_scale=4.0, growth_interval=3)
    n = 1024
    p = torch.ones(2 * n, device=DEV)
    g = torch.full((2 * n,), 0.5, device=DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    scales = []
    for it in range(7):
        s.begin_step()
        if it == 4:
            g[n + 3] = float("inf")
        s.check(0, g[:n]); s.check(1, g[n:])
        ops.adam_step_ranges_amp(p, g, m, v, [(0, n, it + 1, 1e-2), (n, 2 * n, it + 1, 1e-2)], found_inf=s.found_inf, flags=[0, 1], skipped=s.skipped,
                                 lag_index=s.lag_index, count_skip=True)
        s.update()
        g[n + 3] = 0.5
        scales.append(s.get_scale())
    assert scales == [4.0, 4.0, 8.0, 8.0, 4.0, 4.0, 4.0]  # growth after 3 clean steps, backoff at the inf, tracker restarted
    assert s.num_skipped(0) == 0 and s.num_skipped(1) == 1 and s.schedule_lag() == 1
    ref = torch.optim.Adam([torch.nn.Parameter(torch.ones(n, device=DEV))], lr=1e-2, eps=1e-15)
    for it in range(7):
        ref.param_groups[0]["params"][0].grad = torch.full((n,), 0.5, device=DEV)
        ref.step()
    assert float((p[:n] - ref.param_groups[0]["params"][0].detach()).abs().max()) <= 2e-6  # group 0: 7 steps
    ref6 = torch.optim.Adam([torch.nn.Parameter(torch.ones(n, device=DEV))], lr=1e-2, eps=1e-15)
    for it in range(6):
        ref6.param_groups[0]["params"][0].grad = torch.full((n,), 0.5, device=DEV)
        ref6.step()
    assert float((p[n:] - ref6.param_groups[0]["params"][0].detach()).abs().max()) <= 2e-6  # group 1: 6 steps (one skipped)


@pytest.mark.parametrize("single_thread_backward", [False, True])
def test_distributed_data_parallel_wrap_single_rank(golden_dir, single_thread_backward):
    """pipelines/base_pipeline.py:281-283: DDP(model, device_ids=[local_rank], find_unused_parameters=True).  The parameters are views of one
    arena, the proposal networks get None gradients on non-update iterations and the thermal twins never get one in shared mode: the wrapped
    model must take the same steps as the unwrapped one (1-rank RCCL group: the all-reduce is the identity).
    single_thread_backward: the opt-in host setting nerfstudio_thermal_amd.configure_host() -- DDP's reducer hooks then run on the calling thread."""
    import torch.distributed as dist

    import nerfstudio_thermal_amd as pkg
    from torch.nn.parallel import DistributedDataParallel as DDP

    from nerfstudio_thermal_amd.parallel import free_port

    created = False
    if not dist.is_initialized():
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    pkg.configure_host(single_thread_backward=single_thread_backward)
    try:
        mA, oA, rb, batch, _ = _setup(golden_dir, "shared")
        mB, oB, _, _, _ = _setup(golden_dir, "shared")
        ddp = DDP(mA, device_ids=[torch.cuda.current_device()], find_unused_parameters=True)
        sA, sB = torch.amp.GradScaler("cuda"), torch.amp.GradScaler("cuda")
        seen_idle = False
        for step in range(13):
            _sync(mB, oB, mA, oA)
            torch.manual_seed(100 + step)
            mA.engine.__dict__.pop("_rand", None)  # both models draw this iteration's jitter from the same seed
            _reference_train_iteration(mA, oA, sA, rb, batch, step, True, call=ddp)
            torch.manual_seed(100 + step)
            mB.engine.__dict__.pop("_rand", None)
            _reference_train_iteration(mB, oB, sB, rb, batch, step, True, call=mB)
            seen_idle = seen_idle or not mA.engine.last_updated
            assert mA.engine.last_updated == mB.engine.last_updated
            assert_same_training_state(_snapshot(mA), _snapshot(mB), f"DistributedDataParallel-wrapped vs plain model, iteration {step}")
        assert seen_idle
    finally:
        pkg.configure_host(single_thread_backward=False)
        if created:
            dist.destroy_process_group()


def test_load_torch_adam_checkpoint_with_idle_parameters(golden_dir):
    """torch.optim.Adam creates its state lazily: a checkpoint written by it (or by the reference Trainer) has no entry for parameters that never
    received a gradient.  HipFusedAdam.load_state_dict must give those zero moments and step 0 (not stale arena contents) and emit full entries
    that torch.optim.Adam can step from again."""
    from nerfstudio_thermal_amd.optim import HipFusedAdam

    mA, oA, rb, batch, jit = _setup(golden_dir, "shared", HipFusedAdam)
    params = oA.parameters["fields"]
    idle = {id(params[0]), id(params[3])}
    ref = torch.optim.Adam(params, lr=1e-2, eps=1e-15)
    torch.manual_seed(0)
    for _ in range(3):
        for p in params:
            p.grad = None if id(p) in idle else torch.randn_like(p) * 1e-3
        ref.step()
    ckpt = ref.state_dict()
    assert len(ckpt["state"]) == len(params) - 2
    hip = oA.optimizers["fields"]
    mA.arena.exp_avg.fill_(7.0)  # stale contents that must not survive for the idle parameters
    mA.arena.exp_avg_sq.fill_(7.0)
    hip.load_state_dict(ckpt)
    for i, p in enumerate(params):
        st = hip.state[p]
        if id(p) in idle:
            assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0 and float(st["step"]) == 0.0
        else:
            assert torch.equal(st["exp_avg"], ref.state[p]["exp_avg"]) and float(st["step"]) == 3.0
    # round trip: torch.optim.Adam loads what HipFusedAdam writes and both take the same next step
    import copy

    sd = copy.deepcopy(hip.state_dict())  # (the moments in it are views of the arena: freeze them before hip steps)
    assert all(set(v.keys()) == {"step", "exp_avg", "exp_avg_sq"} for v in sd["state"].values()) and len(sd["state"]) == len(params)
    before = [p.detach().clone() for p in params]
    grads = [torch.randn_like(p) * 1e-3 for p in params]
    for p, g in zip(params, grads):
        p.grad = g.clone()
    hip.step()
    after_hip = [p.detach().clone() for p in params]
    with torch.no_grad():
        for p, b in zip(params, before):
            p.copy_(b)
    ref2 = torch.optim.Adam(params, lr=1e-2, eps=1e-15)
    ref2.load_state_dict(sd)
    for p, g in zip(params, grads):
        p.grad = g.clone()
    ref2.step()
    for p, a in zip(params, after_hip):
        assert float((p.detach() - a).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


def test_fused_checkpoint_carries_the_grad_scaler_state(golden_dir):
    """Resume of the fused path after a skipped iteration (reference: Trainer.save_checkpoint stores "scalers": grad_scaler.state_dict(),
    engine/trainer.py:444, restored at :408-419; torch's Adam `step` and the schedulers exclude the skipped iteration): the checkpoint written
    after [clean, clean, inf, clean] carries scale / growth tracker, per-group steps WITHOUT the skipped one and last_epoch WITHOUT the lag, and a
    model resumed from it takes the same next step as the one that kept running."""
    from nerfstudio_thermal_amd.optim import DeviceGradScaler

    mA, _, rb, batch, jit = _setup(golden_dir, "shared")
    rays = (rb.origins.contiguous(), rb.directions.contiguous(), rb.camera_indices.reshape(-1).contiguous())
    sA = DeviceGradScaler(DEV)
    bad = batch["image"].clone()
    bad[:, :] = float("inf")
    for step in range(4):
        mA.engine.train_step(*rays, bad if step == 2 else batch["image"], batch["is_thermal"], step, jit[0], jit[1], grad_scaler=sA)
    ck = {"model": {k: v.clone() for k, v in mA.state_dict().items()}, **mA.engine.optimizer_state_dict(grad_scaler=sA)}
    assert ck["scalers"]["scale"] == 32768.0 and ck["scalers"]["_growth_tracker"] == 1
    # what the reference Trainer would have written for this history: 3 real field steps, 4 proposal steps, 3 scheduler steps
    assert float(ck["optimizers"]["fields"]["state"][0]["step"]) == 3.0
    assert float(ck["optimizers"]["proposal_networks"]["state"][0]["step"]) == 4.0
    assert ck["schedulers"]["fields"]["last_epoch"] == 3
    mB, _, _, _, _ = _setup(golden_dir, "shared")
    sB = DeviceGradScaler(DEV)
    mB.load_model(ck)
    mB.engine.load_optimizer_state_dict(ck, grad_scaler=sB)
    assert sB.get_scale() == 32768.0 and sB.schedule_lag() == 0 and int(sB.growth_tracker.item()) == 1
    assert_same_training_state(_snapshot(mA), _snapshot(mB), "state right after loading the checkpoint", frac=0.0)
    mA.engine.train_step(*rays, batch["image"], batch["is_thermal"], 4, jit[0], jit[1], grad_scaler=sA)
    mB.engine.train_step(*rays, batch["image"], batch["is_thermal"], 4, jit[0], jit[1], grad_scaler=sB)
    # one iteration from identical states: the same bias corrections (step 4 of the fields, 5 of the proposal networks) and the same LR (schedule at 3)
    assert_same_training_state(_snapshot(mA), _snapshot(mB), "first step after the resume")
    # a plain torch.amp.GradScaler state (a reference checkpoint) loads too
    sC = DeviceGradScaler(DEV)
    sC.load_state_dict(torch.amp.GradScaler("cuda", init_scale=1024.0).state_dict())
    assert sC.get_scale() == 1024.0 and sC.schedule_lag() == 0


def test_fused_step_after_a_drop_in_step_starts_from_zero_gradients(golden_dir):
    """The fused step skips its zero-fill when the previous optimiser launch consumed the gradients.  A step on the drop-in path in between
    writes into the same buffer (model._RenderFn.backward, autograd through the aliased .grad views) and its optimiser does not consume: the
    arena's dirty bit must make the next fused step clear the buffer -- fused, drop-in, fused == the same three steps with explicit zero-fills."""
    from nerfstudio_thermal_amd.optim import HipFusedAdam

    mA, oA, rb, batch, jit = _setup(golden_dir, "shared", HipFusedAdam)
    mB, oB, _, _, _ = _setup(golden_dir, "shared", HipFusedAdam)
    rays = (rb.origins.contiguous(), rb.directions.contiguous(), rb.camera_indices.reshape(-1).contiguous())
    for m, force in ((mA, False), (mB, True)):
        opt = oA if m is mA else oB
        m.engine.train_step(*rays, batch["image"], batch["is_thermal"], 0, jit[0], jit[1])
        assert m.arena.grads_clean
        _reference_train_iteration(m, opt, torch.amp.GradScaler("cuda", enabled=False), rb, batch, 1, False, jit)
        assert not m.arena.grads_clean  # the drop-in step left its gradients in the buffer
        if force:
            m.arena.zero_grad()  # what the flag must trigger by itself in run A
        m.engine.train_step(*rays, batch["image"], batch["is_thermal"], 2, jit[0], jit[1])
    assert_same_training_state(_snapshot(mA), _snapshot(mB), "fused / drop-in / fused against the same with an explicit zero-fill")
    # a manual optimiser step that skips a group which HAD gradients leaves the buffer dirty
    mA.engine.train_step(*rays, batch["image"], batch["is_thermal"], 3, jit[0], jit[1])
    out, br = mA.engine.get_outputs(*rays, True, jit[0], jit[1])
    mA.engine.loss_and_backward(out, br, rays[2], batch["image"], batch["is_thermal"])
    mA.engine.optimizer_step(skip_groups=("camera_opt",))
    assert not mA.arena.grads_clean


def test_one_call_train_step_is_the_five_call_sequence(golden_dir, monkeypatch):
    """tn_train_step (RenderEngine._train_step_one_call) against the same iteration through its five library calls (TN_TRAIN_STEP_ONE_CALL=0):
    same rays, same jitter, same parameters -> the same losses (the forward and the loss kernels are deterministic up to the order of the
    float atomics that add the 64 loss lines), the same training state after every iteration (compared from re-synchronised states, as the
    other tests here do), the same sampler / optimiser bookkeeping; and a forced-inf iteration is skipped the same way."""
    import nerfstudio_thermal_amd.engine as E
    from nerfstudio_thermal_amd.optim import DeviceGradScaler

    mA, _, rb, batch, jit = _setup(golden_dir, "shared")
    mB, _, _, _, _ = _setup(golden_dir, "shared")
    rays = (rb.origins.contiguous(), rb.directions.contiguous(), rb.camera_indices.reshape(-1).contiguous())
    sA, sB = DeviceGradScaler(DEV), DeviceGradScaler(DEV)
    bad = batch["image"].clone()
    bad[:, :] = float("inf")
    used = []
    orig = E.RenderEngine._train_step_one_call
    monkeypatch.setattr(E.RenderEngine, "_train_step_one_call", lambda self, *a, **k: (used.append(1), orig(self, *a, **k))[1])
    for step in range(13):  # (covers iterations with and without a proposal update: the schedule thins out after the first ten)
        img = bad if step == 6 else batch["image"]
        mB.arena.params.copy_(mA.arena.params); mB.arena.exp_avg.copy_(mA.arena.exp_avg); mB.arena.exp_avg_sq.copy_(mA.arena.exp_avg_sq)
        monkeypatch.setattr(E, "_ONE_CALL_STEP", True)
        n0 = len(used)
        lA = mA.engine.train_step(*rays, img, batch["is_thermal"], step, jit[0], None, grad_scaler=sA)
        assert len(used) == n0 + 1  # the one-call path ran
        monkeypatch.setattr(E, "_ONE_CALL_STEP", False)
        lB = mB.engine.train_step(*rays, img, batch["is_thermal"], step, jit[0], None, grad_scaler=sB)
        assert len(used) == n0 + 1
        assert lA.keys() == lB.keys()
        if step != 6:
            for k in lA:
                assert abs(float(lA[k]) - float(lB[k])) <= 1e-5 * abs(float(lB[k])) + 1e-12, (step, k, float(lA[k]), float(lB[k]))
        assert_same_training_state(_snapshot(mA), _snapshot(mB), f"tn_train_step vs the five calls, iteration {step}")
        eA, eB = mA.engine, mB.engine
        assert (eA.last_updated, eA.steps_since_update, eA.adam_step_count, eA.group_steps) == (eB.last_updated, eB.steps_since_update, eB.adam_step_count,
                                                                                               eB.group_steps)
        assert sA.get_scale() == sB.get_scale() and sA.schedule_lag() == sB.schedule_lag()
        assert [sA.num_skipped(i) for i in range(3)] == [sB.num_skipped(i) for i in range(3)]
        assert mA.arena.grads_clean and float(mA.arena.grads.abs().max()) == 0.0  # the launch consumed the gradients
    assert sA.get_scale() == 32768.0 and sA.schedule_lag() == 1


def test_one_call_train_step_with_the_fused_renderer_launch(golden_dir, monkeypatch):
    """TN_FUSE_RENDER=1: tn_train_step with tn_render_losses_bwd (renderers + losses + renderer backward in one launch; a measured experiment,
    off by default) against the default tn_train_step: same losses to rounding, the same training state after every iteration -- with and
    without a proposal update, and through a skipped (forced-inf) iteration."""
    from nerfstudio_thermal_amd.optim import DeviceGradScaler

    mA, _, rb, batch, jit = _setup(golden_dir, "shared")
    mB, _, _, _, _ = _setup(golden_dir, "shared")
    rays = (rb.origins.contiguous(), rb.directions.contiguous(), rb.camera_indices.reshape(-1).contiguous())
    sA, sB = DeviceGradScaler(DEV), DeviceGradScaler(DEV)
    bad = batch["image"].clone()
    bad[:, :] = float("inf")
    saw_plain = False
    for step in range(13):
        img = bad if step == 4 else batch["image"]
        mB.arena.params.copy_(mA.arena.params); mB.arena.exp_avg.copy_(mA.arena.exp_avg); mB.arena.exp_avg_sq.copy_(mA.arena.exp_avg_sq)
        monkeypatch.setenv("TN_FUSE_RENDER", "1")
        lA = mA.engine.train_step(*rays, img, batch["is_thermal"], step, jit[0], None, grad_scaler=sA)
        torch.cuda.synchronize()
        monkeypatch.delenv("TN_FUSE_RENDER")
        lB = mB.engine.train_step(*rays, img, batch["is_thermal"], step, jit[0], None, grad_scaler=sB)
        torch.cuda.synchronize()
        saw_plain |= not mA.engine.last_updated
        if step != 4:
            for k in lA:
                assert abs(float(lA[k]) - float(lB[k])) <= 1e-5 * abs(float(lB[k])) + 1e-12, (step, k, float(lA[k]), float(lB[k]))
        assert_same_training_state(_snapshot(mA), _snapshot(mB), f"fused renderer launch vs the three launches, iteration {step}")
        # everything the forward leaves in its buffer (samples, densities, weights, composite, depths) bit for bit -- among it the expected depth
        # after the batch-wide clip, which the default path runs as co-work blocks of the loss launch and the fused path as a launch of its own
        cA, cB = mA.engine._step_call, mB.engine._step_call
        assert cA.off == cB.off
        n = rays[0].shape[0]
        S0, S1, S2 = cA.counts
        # slots of tn_render_rays_train_layout in order (the regions are padded to 256 bytes: only their contents are compared)
        sizes = [n * 3, n * 3, n * (S0 + 1), n * (S0 + 1), n * S0, n * S0, n, n * (S1 + 1), n * (S1 + 1), n * S1, n * S1, n, n * (S2 + 1), n * (S2 + 1), n * S2,
                 n * S2, n * S2 * 4, n * 4, n, n, n]
        for slot, size in enumerate(sizes):
            o = cA.off[slot]
            assert torch.equal(cA._keep[3][o:o + size], cB._keep[3][o:o + size]), (step, slot)
        assert sA.get_scale() == sB.get_scale() and [sA.num_skipped(i) for i in range(3)] == [sB.num_skipped(i) for i in range(3)]
    assert saw_plain  # an iteration without a proposal update was among them

