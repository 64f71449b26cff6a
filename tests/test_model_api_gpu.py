"""The drop-in boundary, tier 2: ThermalNerfactoModel with the reference's Model API on the GPU.
state_dict contract, forward/get_outputs keys and shapes, autograd-compatible training step, fused training step, chunked camera render."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import golden_inputs, make_params, tiny_cfg
from test_hip_ops_gpu import md, pkg_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_model(mode):
    from nerfstudio_thermal_amd.model import SceneBox

    ocfg = tiny_cfg(mode)
    cfg = pkg_cfg(ocfg)
    model = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=ocfg.num_images,
                      metadata={"is_thermal": list(ocfg.is_thermal_cam)}, device=DEV)
    return ocfg, cfg, model


def bundle(golden_dir):
    from nerfstudio_thermal_amd.rays import RayBundle

    gi = golden_inputs(golden_dir)
    g = np.load(os.path.join(golden_dir, "raygen.npz"))
    rb = RayBundle(origins=gi["origins"].to(DEV), directions=gi["directions"].to(DEV), pixel_area=torch.from_numpy(g["pixel_area"]).to(DEV),
                   camera_indices=gi["camera_indices"][:, None].to(DEV))
    return gi, rb


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_state_dict_contract_and_checkpoint_roundtrip(golden_dir, mode):
    ocfg, cfg, model = build_model(mode)
    ref = json.load(open(os.path.join(golden_dir, f"state_dict_keys_{mode}.json")))
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(ref.keys())
    for k, (shape, dtype) in ref.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype, (k, sd[k].shape, shape, sd[k].dtype, dtype)
    # aliased proposal tables stay aliased and every parameter aliases the flat arena
    a = model.proposal_networks[0]
    assert a.encoding.hash_table.data_ptr() == a.mlp_base._modules["0"].hash_table.data_ptr()
    lo, hi = model.arena.params.data_ptr(), model.arena.params.data_ptr() + model.arena.params.numel() * 4
    for n, p in model.named_parameters():
        if n != "device_indicator_param":
            assert lo <= p.data_ptr() < hi, n
    # load a reference-style (DDP-prefixed) checkpoint in place: the arena must see the new values
    params = make_params(ocfg)
    state = {"module." + k: v for k, v in sd.items()}
    for k, v in params.items():
        state["module." + k] = v
        alias = k.replace("mlp_base.0.hash_table", "encoding.hash_table")
        if alias != k:
            state["module." + alias] = v
    model.load_model({"model": state})
    for k, v in params.items():
        assert md(model.arena.view(k), v) == 0.0, k
    assert sorted(model.get_param_groups().keys()) == sorted(
        ["proposal_networks", "fields", "camera_opt"] + (["proposal_networks_thermal", "fields_thermal", "camera_opt_thermal"] if mode == "separate" else []))


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_forward_eval_keys_shapes_values(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, f"model_{mode}.npz"))
    ocfg, cfg, model = build_model(mode)
    model.arena.load(make_params(ocfg))
    gi, rb = bundle(golden_dir)
    model.eval()
    with torch.no_grad():
        out = model(rb)
    ref_keys = sorted(k[5:] for k in g.files if k.startswith("eval/"))
    assert sorted(k for k, v in out.items() if isinstance(v, torch.Tensor)) == ref_keys
    for k in ref_keys:
        assert tuple(out[k].shape) == g[f"eval/{k}"].shape, k
    assert md(out["rgb"], g["eval/rgb"]) <= 1e-3 and md(out["rgb_thermal"], g["eval/rgb_thermal"]) <= 1e-3
    assert rb.nears is not None and float(rb.nears.max()) == 0.0 and float(rb.fars.min()) == 1000.0  # collider mutated the bundle (eval: near reset to 0)


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_autograd_training_step_matches_fused_step(golden_dir, mode):
    """Trainer-style step (loss_dict -> backward -> param.grad) against the fused no-tape step on the same inputs."""
    g = np.load(os.path.join(golden_dir, f"model_{mode}.npz"))
    gi, rb = bundle(golden_dir)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]
    batch = {"image": gi["image"].to(DEV), "is_thermal": gi["is_thermal"].to(DEV)}
    # --- autograd path
    ocfg, cfg, model = build_model(mode)
    model.arena.load(make_params(ocfg))
    model.train()
    for cb in model.get_training_callbacks():
        cb.run_callback_at_location(500, list(cb.where_to_run)[0]) if "BEFORE" in cb.where_to_run[0].name else None
    assert abs(model.engine.anneal - float(g["train/anneal"])) < 1e-12
    rb1 = model.collider(rb[...])
    out = model.get_outputs(rb1, jit, jit_t)
    metrics = model.get_metrics_dict(out, batch)
    losses = model.get_loss_dict(out, batch, metrics)
    ref_keys = sorted(k[5:] for k in g.files if k.startswith("loss/") and k != "loss/total")
    assert sorted(losses.keys()) == ref_keys
    for k in ref_keys:
        a, b = float(losses[k]), float(g[f"loss/{k}"])
        assert abs(a - b) <= 2e-4 * abs(b) + 1e-9, (k, a, b)
    sum(losses.values()).backward()
    grads_auto = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    # --- fused path on a fresh model
    ocfg2, cfg2, model2 = build_model(mode)
    model2.arena.load(make_params(ocfg2))
    model2.train()
    eng = model2.engine
    eng.set_anneal_for_step(500)
    eng.arena.zero_grad()
    o, d, cam = rb.origins.contiguous(), rb.directions.contiguous(), rb.camera_indices.reshape(-1).contiguous()
    out2, br2 = eng.get_outputs(o, d, cam, True, jit, jit_t)
    eng.loss_and_backward(out2, br2, cam, batch["image"], batch["is_thermal"])
    for name in model2.arena.names():
        fused = model2.arena.grad_view(name)
        key = name if name in grads_auto else name.replace("mlp_base.0.hash_table", "encoding.hash_table")
        if float(fused.abs().max()) == 0.0:
            assert key not in grads_auto or float(grads_auto[key].abs().max()) == 0.0, name
            continue
        scale = float(fused.abs().max())
        assert md(grads_auto[key], fused) <= 1e-4 * scale, (name, md(grads_auto[key], fused), scale)


def test_train_iteration_reduces_loss_and_camera_render_chunks():
    from nerfstudio_thermal_amd import ops, synth
    from nerfstudio_thermal_amd.rays import RayBundle

    ocfg, cfg, model = build_model("shared")
    cfg.eval_num_rays_per_chunk = 1000  # force several chunks
    cams = synth.synth_cameras()
    N = 1024
    idx = torch.from_numpy(synth.synth_ray_indices(cams, N)).to(DEV)
    t = lambda k: torch.from_numpy(cams[k]).to(DEV)  # noqa: E731
    o, d, area, _ = ops.raygen(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
    img, is_th = (torch.from_numpy(a).to(DEV) for a in synth.synth_gt(idx.cpu().numpy(), cams))
    img = torch.where(is_th[:, None] > 0, torch.full_like(img, 0.7), torch.tensor([0.2, 0.5, 0.8], device=DEV).expand_as(img)).contiguous()
    model.train()
    first = last = None
    for step in range(60):
        rb = RayBundle(origins=o, directions=d, pixel_area=area, camera_indices=idx[:, 0:1].contiguous())
        losses = model.train_iteration(rb, {"image": img, "is_thermal": is_th}, step)
        tot = float(losses["rgb_loss"] + losses["thermal_loss"])
        first = tot if first is None else first
        last = tot
    assert np.isfinite(last) and last < 0.25 * first, (first, last)
    # full-"image" render: a 40 x 30 ray grid through the chunked eval path
    model.eval()
    H, W = 20, 40
    rb = RayBundle(origins=o[: H * W].reshape(H, W, 3), directions=d[: H * W].reshape(H, W, 3), pixel_area=area[: H * W].reshape(H, W, 1),
                   camera_indices=idx[: H * W, 0:1].reshape(H, W, 1).contiguous())
    outs = model.get_outputs_for_camera_ray_bundle(rb)
    assert outs["rgb"].shape == (H, W, 3) and outs["rgb_thermal"].shape == (H, W, 1) and outs["density"].shape == (H, W, 48)
    assert bool(torch.isfinite(outs["rgb"]).all())


def test_get_outputs_for_camera_and_image_metrics():
    """Model.get_outputs_for_camera (models/base_model.py:165-175) on a thermal 160x120 camera == the oracle's eval render of the same pixels,
    and get_image_metrics_and_images (models/thermal_nerfacto.py:490-564) on it: PSNR against an image we control, SSIM sanity."""
    import types

    import thermal_nerfacto_oracle as orc
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.model import _ssim

    ocfg, cfg, model = build_model("shared")
    model.eval()
    cams = synth.synth_cameras()
    c = 4  # thermal camera
    cam = types.SimpleNamespace(camera_to_worlds=torch.from_numpy(cams["c2w"][c]), fx=float(cams["fx"][c]), fy=float(cams["fy"][c]), cx=float(cams["cx"][c]),
                                cy=float(cams["cy"][c]), width=int(cams["width"][c]), height=int(cams["height"][c]),
                                distortion_params=torch.from_numpy(cams["distortion"][c]), camera_index=c)
    outs = model.get_outputs_for_camera(cam)
    H, W = int(cams["height"][c]), int(cams["width"][c])
    assert outs["rgb"].shape == (H, W, 3) and outs["rgb_thermal"].shape == (H, W, 1) and outs["depth"].shape == (H, W, 1)
    # the oracle on a strip of the same image (rows 50..53)
    yy, xx = np.meshgrid(np.arange(50, 54), np.arange(W), indexing="ij")
    idx = torch.from_numpy(np.stack([np.full(yy.size, c), yy.reshape(-1), xx.reshape(-1)], 1).astype(np.int64))
    t = lambda k: torch.from_numpy(cams[k])  # noqa: E731
    ro, rd, _, _ = orc.generate_rays(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
    params = {k: v.detach().cpu() for k, v in model.state_dict().items() if k in orc.param_shapes(ocfg)}
    with torch.no_grad():
        ref = orc.get_outputs(params, ocfg, ro, rd, idx[:, 0], training=False)
    assert md(outs["rgb_thermal"][50:54].reshape(-1, 1), ref["rgb_thermal"]) <= 1e-3
    assert md(outs["rgb"][50:54].reshape(-1, 3), ref["rgb"]) <= 1e-3
    # metrics: ground truth = the prediction shifted by a constant 0.1 -> PSNR = 20 dB exactly, SSIM < 1; identical image -> SSIM = 1
    # (an image with real structure as the "prediction": the freshly initialised model renders an almost constant image, for which SSIM with a
    # data-derived range is numerically meaningless -- in torchmetrics as well)
    pred_th = (torch.from_numpy(synth.synth_images(cams)[c][..., :1]).to(DEV) * 0.8).contiguous()
    outs["rgb_thermal"] = pred_th
    gt = (pred_th + 0.1).expand(-1, -1, 3).contiguous()
    metrics, images = model.get_image_metrics_and_images(outs, {"image": gt, "is_thermal": 1})
    assert set(metrics) == {"psnr_thermal", "ssim_thermal"} and abs(metrics["psnr_thermal"] - 20.0) < 1e-3 and 0.0 < metrics["ssim_thermal"] < 1.0
    assert images["img"].shape == (H, 3 * W, 3) and set(images) >= {"img", "accumulation", "depth", "prop_depth_0", "prop_depth_1"}
    x = torch.moveaxis(pred_th, -1, 0)[None]
    assert abs(float(_ssim(x, x)) - 1.0) < 1e-6


# ------------------------------------------------------------------------------------------------ the Trainer-style path, end to end
def _trainer_step(model, optimizers, rb, batch, step, jit=None, jit_t=None):
    """Trainer.train_iteration (engine/trainer.py:455-499) with this package's pieces: callbacks, zero_grad, forward, metrics, loss dict,
    backward, optimiser + scheduler step."""
    from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc

    cbs = model.get_training_callbacks()
    for cb in cbs:
        cb.run_callback_at_location(step, Loc.BEFORE_TRAIN_ITERATION)
    optimizers.zero_grad_all()
    out = model.get_outputs(model.collider(rb[...]), jit, jit_t)
    metrics = model.get_metrics_dict(out, batch)
    losses = model.get_loss_dict(out, batch, metrics)
    sum(losses.values()).backward()
    optimizers.optimizer_step_all(step)
    optimizers.scheduler_step_all(step)
    for cb in cbs:
        cb.run_callback_at_location(step, Loc.AFTER_TRAIN_ITERATION)
    return losses


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_parameters_without_gradient_get_none(golden_dir, mode):
    """On iterations where the sampler does not update the proposal networks they run under no_grad in the reference
    (model_components/ray_samplers.py:591,605-610): .grad stays None and torch.optim.Adam skips them.  Same here, and in shared mode the
    thermal twins never receive a gradient."""
    from nerfstudio_thermal_amd.optim import Optimizers

    gi, rb = bundle(golden_dir)
    batch = {"image": gi["image"].to(DEV), "is_thermal": gi["is_thermal"].to(DEV)}
    ocfg, cfg, model = build_model(mode)
    model.arena.load(make_params(ocfg))
    model.train()
    opt = Optimizers(model.get_param_groups())
    seen_idle = False
    for step in range(14):
        before = {n: p.detach().clone() for n, p in model._params.items()}
        _trainer_step(model, opt, rb, batch, step)
        updated = model.engine.last_updated
        for n, p in model._params.items():
            is_prop = n.startswith("proposal_networks.")
            thermal_twin = "_thermal" in n
            if (is_prop and not updated) or (thermal_twin and mode == "shared"):
                assert p.grad is None, (step, n)
                assert torch.equal(p.detach(), before[n]), (step, n)  # not stepped, no coasting on momentum
            else:
                assert p.grad is not None, (step, n)
        seen_idle = seen_idle or not updated
    assert seen_idle  # from step 10 the schedule skips iterations
    # per-parameter Adam step counts follow the gradients: the proposal networks have fewer steps than the field
    k_prop = {int(v["step"]) for v in opt.optimizers["proposal_networks"].state_dict()["state"].values()}
    k_field = {int(v["step"]) for v in opt.optimizers["fields"].state_dict()["state"].values()}
    assert k_field == {14} and len(k_prop) == 1 and k_prop.pop() < 14


def test_hip_fused_adam_matches_torch_adam(golden_dir):
    """HipFusedAdam (one kernel launch per group over the arena) against torch.optim.Adam: the same gradients for 5 steps, then one real iteration."""
    from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers

    gi, rb = bundle(golden_dir)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    batch = {"image": gi["image"].to(DEV), "is_thermal": gi["is_thermal"].to(DEV)}
    # (1) the update rule, deterministically: the same synthetic gradients through both optimisers for 5 steps -- the proposal networks
    # without a gradient on steps 2 and 3 (skipped by both, their step counts fall behind the field's) -- must give the same parameters.
    # (Comparing whole training runs instead is ill-posed: the float-atomic sums of the backward differ in the last bits from run to run and
    # Adam with eps = 1e-15 turns a sign flip of a ~1e-12 gradient into a 2 lr difference.)
    frac = lambda a, b: float(((a - b).abs() > 1e-5).float().mean())  # noqa: E731
    models, opts = [], []
    for cls in (HipFusedAdam, torch.optim.Adam):
        ocfg, cfg, model = build_model("shared")
        model.arena.load(make_params(ocfg))
        model.train()
        models.append(model)
        opts.append(Optimizers(model.get_param_groups(), optimizer_cls=cls))
    gen = torch.Generator(device=DEV).manual_seed(7)
    for step in range(5):
        grads = {}
        for n, p in models[0]._params.items():
            idle = "_thermal" in n or (n.startswith("proposal_networks.") and step in (2, 3))
            # a sparse gradient with a wide dynamic range (exact zeros included, as the hash tables have)
            g = torch.randn(p.shape, device=DEV, generator=gen) * torch.pow(10.0, torch.randint(-8, 1, p.shape, device=DEV, generator=gen).float())
            grads[n] = None if idle else g * (torch.rand(p.shape, device=DEV, generator=gen) < 0.5)
        for model, opt in zip(models, opts):
            opt.zero_grad_all()
            for n, p in model._params.items():
                if grads[n] is not None:
                    view = model.arena.grad_view(n)
                    view.copy_(grads[n])
                    p.grad = view
            opt.optimizer_step_all(step)
            opt.scheduler_step_all(step)
    for n, p in models[0]._params.items():
        q = models[1]._params[n]
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-6, (n, float((p.detach() - q.detach()).abs().max()))
    del models, opts
    # and on a single iteration from identical state the two are the same update (up to the same noise on near-zero gradients)
    one = []
    for cls in (HipFusedAdam, torch.optim.Adam):
        ocfg, cfg, model = build_model("shared")
        model.arena.load(make_params(ocfg))
        model.train()
        opt = Optimizers(model.get_param_groups(), optimizer_cls=cls)
        _trainer_step(model, opt, rb, batch, 0, jit)
        one.append({n: p.detach().clone() for n, p in model._params.items()})
    for n in one[0]:
        assert frac(one[0][n], one[1][n]) <= 0.01, (n, frac(one[0][n], one[1][n]))


def test_gradient_accumulation_two_micro_steps(golden_dir):
    """Trainer with gradient_accumulation_steps = 2 (engine/trainer.py:464-477): two forward/backward passes before the optimiser step must
    leave grad(batch A) + grad(batch B) in param.grad."""
    gi, rb = bundle(golden_dir)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    batch = {"image": gi["image"].to(DEV), "is_thermal": gi["is_thermal"].to(DEV)}
    batch_b = {"image": (1.0 - gi["image"]).to(DEV), "is_thermal": gi["is_thermal"].to(DEV)}

    def grads_of(batches):
        ocfg, cfg, model = build_model("shared")
        model.arena.load(make_params(ocfg))
        model.train()
        for b in batches:
            out = model.get_outputs(model.collider(rb[...]), jit, None)
            losses = model.get_loss_dict(out, b, model.get_metrics_dict(out, b))
            sum(losses.values()).backward()
        return {n: p.grad.detach().clone() for n, p in model._params.items() if p.grad is not None}

    ga, gb, gab = grads_of([batch]), grads_of([batch_b]), grads_of([batch, batch_b])
    for n in ga:
        ref = ga[n] + gb[n]
        assert md(gab[n], ref) <= 1e-4 * max(float(ref.abs().max()), 1e-12), n


def test_optimizer_state_roundtrip_fused_and_trainer_paths(golden_dir):
    """Resume: 3 + 2 iterations through a checkpoint == 5 iterations straight, for the fused engine path (engine.optimizer_state_dict, the
    reference's "optimizers"/"schedulers" layout) and for HipFusedAdam's torch-format state_dict."""
    from nerfstudio_thermal_amd.optim import Optimizers

    gi, rb = bundle(golden_dir)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    batch = {"image": gi["image"].to(DEV), "is_thermal": gi["is_thermal"].to(DEV)}
    o, d, cam = rb.origins.contiguous(), rb.directions.contiguous(), rb.camera_indices.reshape(-1).contiguous()

    def fresh():
        ocfg, cfg, model = build_model("shared")
        model.arena.load(make_params(ocfg))
        model.train()
        return model

    # --- fused path
    def fused_steps(model, steps):
        for s in steps:
            model.engine.train_step(o, d, cam, batch["image"], batch["is_thermal"], s, jit)

    frac = lambda a, b: float(((a - b).abs() > 1e-5).float().mean())  # noqa: E731
    m0 = fresh(); fused_steps(m0, range(5))  # a second straight run: the run-to-run noise of float-atomic sums amplified by Adam
    m1 = fresh(); fused_steps(m1, range(5))
    m2 = fresh(); fused_steps(m2, range(3))
    ckpt = {"model": {k: v.clone() for k, v in m2.state_dict().items()}, **m2.engine.optimizer_state_dict()}
    assert set(ckpt["optimizers"]) == {"proposal_networks", "fields", "camera_opt"}
    assert set(ckpt["optimizers"]["fields"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    m3 = fresh(); m3.load_model(ckpt); m3.engine.load_optimizer_state_dict(ckpt)
    assert m3.engine.adam_step_count == 3 and m3.engine.group_steps["fields"] == 3
    fused_steps(m3, range(3, 5))
    for n in m1.arena.names():
        noise = frac(m0.arena.view(n), m1.arena.view(n))
        # (tensors of a few dozen entries: one noise-driven sign flip is already percents; bound the size of the difference instead)
        small = m1.arena.view(n).numel() < 1000 and float((m1.arena.view(n) - m3.arena.view(n)).abs().max()) <= 0.03
        assert small or frac(m1.arena.view(n), m3.arena.view(n)) <= 2.0 * noise + 0.05, (n, noise)
    # --- Trainer path with HipFusedAdam
    t0 = fresh(); o0 = Optimizers(t0.get_param_groups())
    t1 = fresh(); o1 = Optimizers(t1.get_param_groups())
    for s in range(5):
        _trainer_step(t0, o0, rb, batch, s, jit)
        _trainer_step(t1, o1, rb, batch, s, jit)
    t2 = fresh(); o2 = Optimizers(t2.get_param_groups())
    for s in range(3):
        _trainer_step(t2, o2, rb, batch, s, jit)
    sd_model, sd_opt = {k: v.clone() for k, v in t2.state_dict().items()}, o2.state_dict()
    eng_state = t2.engine.optimizer_state_dict()["sampler"]
    t3 = fresh(); o3 = Optimizers(t3.get_param_groups())
    t3.load_model({"model": sd_model}); o3.load_state_dict(sd_opt)
    t3.engine.steps_since_update, t3.engine.sampler_step = eng_state["steps_since_update"], eng_state["step"]
    for s in range(3, 5):
        _trainer_step(t3, o3, rb, batch, s, jit)
    for n in t1.arena.names():
        noise = frac(t0.arena.view(n), t1.arena.view(n))
        small = t1.arena.view(n).numel() < 1000 and float((t1.arena.view(n) - t3.arena.view(n)).abs().max()) <= 0.03
        assert small or frac(t1.arena.view(n), t3.arena.view(n)) <= 2.0 * noise + 0.05, (n, noise)


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_backward_of_a_subset_of_loss_terms_is_refused(golden_dir, mode):
    """The loss kernels produce the gradient of the SUM of all terms (one pass, shared gradient buffers).  `loss_dict["rgb_loss"].backward()` or a
    sum over a subset gives the other terms weight 0: the reference's per-term autograd handles that, this node cannot -- it must refuse, never
    answer with the gradient of every term.  The sum over the whole dict, scaled by any common factor (GradScaler), is accepted."""
    gi, rb = bundle(golden_dir)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]
    batch = {"image": gi["image"].to(DEV), "is_thermal": gi["is_thermal"].to(DEV)}
    ocfg, cfg, model = build_model(mode)
    model.arena.load(make_params(ocfg))
    model.train()

    def losses():
        out = model.get_outputs(model.collider(rb[...]), jit, jit_t)
        return model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch))

    with pytest.raises(RuntimeError, match="subset of the loss terms"):
        losses()["rgb_loss"].backward()
    with pytest.raises(RuntimeError, match="subset of the loss terms"):
        L = losses()
        (L["rgb_loss"] + L["interlevel_loss"]).backward()
    with pytest.raises(RuntimeError, match="one common weight"):
        L = losses()
        (2.0 * L["rgb_loss"] + sum(v for k, v in L.items() if k != "rgb_loss")).backward()
    # whole dict, common scale 128 (what GradScaler does): gradients = 128 x the unscaled ones
    model.arena.zero_grad()
    for p in model.parameters():
        p.grad = None
    sum(losses().values()).backward()
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        p.grad = None
    model.arena.zero_grad()
    (sum(losses().values()) * 128.0).backward()
    for n, p in model.named_parameters():
        if n in g1:
            assert md(p.grad, g1[n] * 128.0) <= 2e-5 * float(g1[n].abs().max()) * 128.0 + 1e-30, n
