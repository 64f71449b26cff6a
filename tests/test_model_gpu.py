"""GPU parity of the whole path against the committed golden vectors produced by the reference itself
(tests/golden/model_{shared,separate}.npz): eval render, train-mode forward with injected jitter, every loss_dict entry,
parameter gradients, and one Adam step.  Tolerances: 1e-3 abs RGB/thermal, 1e-4 abs density (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_file, golden_inputs, make_params, oracle_density_sensitivity, sample_indices, size_cfg, tiny_cfg
from nerfstudio_thermal_amd import synth
from nerfstudio_thermal_amd.arena import ParamArena
import thermal_nerfacto_oracle as orc
from nerfstudio_thermal_amd.engine import RenderEngine
from test_hip_ops_gpu import md, outlier_fraction, pkg_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(mode, size="tiny"):
    ocfg = size_cfg(size, mode)
    cfg = pkg_cfg(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load(make_params(ocfg))
    return ocfg, cfg, arena, RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))


def dev_inputs(golden_dir, size="tiny"):
    gi = golden_inputs(golden_dir, size)
    to = lambda t: t.to(DEV).contiguous()  # noqa: E731
    return gi, to(gi["origins"]), to(gi["directions"]), to(gi["camera_indices"])


RGB_TOL, DENS_TOL, DEPTH_TOL = 1e-3, 1e-4, 1e-4


K_CHAIN = 6.0       # max |d density| of the chain <= K_CHAIN x the reference's own response to ONE ulp of its field sample bins
U_CHAIN = 4         # share of samples above 1e-4 <= 1.5 x the reference's own share when EVERY field bin moves by U_CHAIN ulps


def chain_sens(golden_dir, mode, size):
    """the two oracle-only measurements the chained bound is derived from (helpers.oracle_density_sensitivity; no GPU involved)"""
    return oracle_density_sensitivity(golden_dir, mode, size, 1), oracle_density_sensitivity(golden_dir, mode, size, U_CHAIN)


def assert_density_chain(got, ref, what, sens1, sensU):
    """Density after the WHOLE chain (two PDF resamplings feed the sample positions).  With the deliberately high-variance synthetic tables the
    reference's own density moves by S1 = 1.2-1.5e-4 when its field sample bins move by ONE fp32 ulp (tests/test_conditioning_cpu.py measures it
    on the oracle alone) and, when every bin moves by 4 ulps, by up to 3.2e-4 with 0.5-2.8 % of the samples above 1e-4 -- a chained max-abs of
    1e-4 is below what the arithmetic defines.  The HIP sampler's field bins sit at a median of 1-2 ulps from the reference's, 85 % within 4,
    95 % within 8 (test_sampler_bins_ulp_distance prints the histogram), so the chain is held to
        max |err| <= K_CHAIN x S1        and        share(|err| > 1e-4) <= 1.5 x the oracle's share under +-U_CHAIN ulps;
    the strict 1e-4 bound is asserted where it is well posed -- on identical sample positions (test_density_on_identical_samples below and
    tests/test_hip_ops_gpu.py::test_field_fwd_bwd)."""
    err = (got.detach().cpu().double() - torch.as_tensor(ref).double()).abs()
    frac = float((err > DENS_TOL).double().mean())
    assert float(err.max()) <= K_CHAIN * sens1["max"], (what, float(err.max()), sens1)
    assert frac <= 1.5 * max(sensU["frac"], 2.0 / err.numel()), (what, frac, sensU)


@pytest.mark.parametrize("size", ["tiny", "default", "default256"])  # "default": the reference at 16 x 2^19 / 5 x 2^17 tables, 64 rays
@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_eval_render_matches_reference_golden(golden_dir, mode, size):
    g = golden_file(golden_dir, mode, size)
    _, _, _, eng = build(mode, size)
    _, o, d, cam = dev_inputs(golden_dir, size)
    out, _ = eng.get_outputs(o, d, cam, training=False)
    checks = {"rgb": RGB_TOL, "rgb_thermal": RGB_TOL, "accumulation": 1e-4, "expected_depth": DEPTH_TOL, "density": None}
    if mode == "shared":
        checks["rgbt"] = RGB_TOL
    else:
        checks.update({"accumulation_thermal": 1e-4, "expected_depth_thermal": DEPTH_TOL, "density_thermal": None, "density2": None,
                       "density2_thermal": None, "removal": RGB_TOL, "removal_thermal": RGB_TOL})
    for k, tol in checks.items():
        ref = g[f"eval/{k}"]
        assert tuple(out[k].shape) == ref.shape, (k, tuple(out[k].shape), ref.shape)
        if tol is None:
            s1, su = chain_sens(golden_dir, mode, size)
            # density2 / density2_thermal: a field on the OTHER branch's samples -- bounded by the larger of the two branches' responses
            pick = (lambda sd: sd["_thermal" if k == "density_thermal" else ""]) if k in ("density", "density_thermal") else (
                lambda sd: {"max": max(v["max"] for v in sd.values()), "frac": max(v["frac"] for v in sd.values())})
            assert_density_chain(out[k], ref, k, pick(s1), pick(su))
        else:
            assert md(out[k], ref) <= tol, (k, md(out[k], ref))
    # discontinuous outputs (searchsorted at 0.5): allow a small fraction of rays to land on the neighbouring sample
    for k in ["depth", "prop_depth_0", "prop_depth_1"] + (["depth_thermal", "prop_depth_0_thermal", "prop_depth_1_thermal"] if mode == "separate" else []):
        assert outlier_fraction(out[k], g[f"eval/{k}"], 1e-5) <= 0.07, k  # <= 2 of 32 rays


@pytest.mark.parametrize("size", ["tiny", "default", "default256"])  # "default": the reference at 16 x 2^19 / 5 x 2^17 tables, 64 rays
@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_train_step_matches_reference_golden(golden_dir, mode, size):
    g = golden_file(golden_dir, mode, size)
    ocfg, cfg, arena, eng = build(mode, size)
    gi, o, d, cam = dev_inputs(golden_dir, size)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]
    eng.set_anneal_for_step(500)
    assert abs(eng.anneal - float(g["train/anneal"])) < 1e-12
    arena.zero_grad()
    out, branches = eng.get_outputs(o, d, cam, True, jit, jit_t)
    sfx = ("", "_thermal") if mode == "separate" else ("",)
    for s in sfx:
        lv = branches[s].levels
        for i in range(3):
            assert outlier_fraction(lv[i].s_bins, g[f"train/sbins{s}_{i}"], 2e-6) <= 0.01, (s, i)
            assert outlier_fraction(lv[i].weights, g[f"train/weights{s}_{i}"], 1e-5) <= 0.01, (s, i)
        s1, su = chain_sens(golden_dir, mode, size)
        assert_density_chain(out[f"density{s}"], g[f"train/density{s}"], f"density{s}", s1[s], su[s])
        assert md(out[f"rgb{s}"], g[f"train/rgb{s}"]) <= RGB_TOL
        assert md(out[f"accumulation{s}"], g[f"train/accumulation{s}"]) <= 1e-4
    losses = eng.loss_and_backward(out, branches, cam, gi["image"].to(DEV), gi["is_thermal"].to(DEV))
    ref_keys = sorted(k[5:] for k in g.files if k.startswith("loss/") and k != "loss/total")
    assert sorted(losses.keys()) == ref_keys
    for k in ref_keys:
        a, b = float(losses[k]), float(g[f"loss/{k}"])
        assert abs(a - b) <= 2e-4 * abs(b) + 1e-9, (k, a, b)
    # Chained gradients are ill-conditioned (helpers.oracle_grad_sensitivity, scripts/diag_grad_conditioning.py): the reference's own
    # main-field / pose gradients move by up to ~3% of their max entry when the sampler input moves by ONE fp32 ulp, because a sample that
    # crosses a fine-level cell boundary changes a piecewise-constant derivative.  So the chain is held to robust statistics here (norm,
    # 95th percentile, bounded max); strict per-entry bounds are asserted on identical sample positions in tests/test_hip_ops_gpu.py.
    bad = []
    for name in arena.names():
        got = arena.grad_view(name).reshape(-1)
        if f"grad_none/{name}" in g.files:
            assert float(got.abs().max()) == 0.0, name
            continue
        ref_norm = float(g[f"grad_norm/{name}"])
        ii = torch.from_numpy(g[f"grad_idx/{name}"]).to(DEV)
        ref = torch.from_numpy(g[f"grad_val/{name}"])
        scale = max(float(ref.abs().max()), 1e-12)
        e_norm = abs(float(got.double().norm()) - ref_norm) / max(ref_norm, 1e-12)
        diff = (got[ii].detach().cpu().double() - ref.double()).abs() / scale
        e_max, e_p95 = float(diff.max()), float(torch.quantile(diff, 0.95))
        # (the 48-entry pose tensors sum d position over every ray of the batch: the few samples that change cell under a 1-ulp shift move
        # their norm by up to 1 % -- helpers.oracle_grad_sensitivity measures 3 % on the reference itself; d position is pinned strictly, per
        # ray, on identical samples in tests/test_fullsize_parity_gpu.py)
        norm_tol = 5e-3 if diff.numel() >= 1000 else 1e-2
        if e_norm > norm_tol or (diff.numel() >= 1000 and e_p95 > 2e-3) or e_max > 5e-2:  # p95 is meaningless on the 48-entry pose tensors
            bad.append((name, e_norm, e_p95, e_max))
    assert not bad, bad
    eng.optimizer_step(scheduled=False)
    for name in arena.names():
        ii = torch.from_numpy(sample_indices(name, arena.view(name).numel())).to(DEV)
        ref = torch.from_numpy(g[f"adam_val/{name}"])
        # At step 1 Adam moves every entry with a non-zero gradient by ~lr*sign(g) (eps=1e-15): an entry whose gradient is pure rounding
        # noise (contributions that cancel to 0 in one summation order and to 1e-12 in another) legitimately differs by lr.  The Adam
        # kernel itself is pinned against torch.optim.Adam in tests/test_hip_ops_gpu.py; here: >= 99% of entries agree, none is off by > 2 lr.
        diff = (arena.view(name).reshape(-1)[ii].detach().cpu().double() - ref.double()).abs()
        # (a 48-entry pose tensor: one such entry is already 2 %)
        assert float((diff > 2e-5).double().mean()) <= max(0.01, 1.0 / diff.numel() + 1e-9), (name, float((diff > 2e-5).double().mean()))
        assert float(diff.max()) <= 2.1e-2, (name, float(diff.max()))


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_fused_launches_equal_one_launch_per_seam(golden_dir, mode, monkeypatch):
    """The engine's hot path (tn_render_rays_train, tn_train_losses + tn_losses_finish, tn_render_bwd, tn_pose_bwd_finish) against the same
    step made of one entry point per reference seam (TN_FUSE_SMALL=0: tn_pose_apply_fwd, tn_spaced_bins, ..., tn_weights_fwd, tn_composite_fwd,
    tn_clip_depth, tn_pixel_losses, tn_proposal_losses, tn_composite_bwd, tn_weights_bwd, tn_pose_apply_bwd, tn_camera_reg): forward tensors
    bit for bit, loss terms and every parameter gradient up to the order of float additions."""
    from nerfstudio_thermal_amd import engine as engine_mod

    gi, o, d, cam = dev_inputs(golden_dir, "tiny")
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(engine_mod, "_FUSE", fuse)
        ocfg, cfg, arena, eng = build(mode, "tiny")
        eng.set_anneal_for_step(500)
        arena.zero_grad()
        out, branches = eng.get_outputs(o, d, cam, True, jit, jit_t)
        losses = eng.loss_and_backward(out, branches, cam, gi["image"].to(DEV), gi["is_thermal"].to(DEV))
        torch.cuda.synchronize()
        res[fuse] = ({k: v.clone() for k, v in out.items() if torch.is_tensor(v)}, {k: float(v) for k, v in losses.items()}, arena.grads.clone())
    (out_a, loss_a, grad_a), (out_b, loss_b, grad_b) = res[True], res[False]
    assert sorted(out_a) == sorted(out_b)
    for k in out_a:
        assert torch.equal(out_a[k].nan_to_num(nan=-7.0), out_b[k].nan_to_num(nan=-7.0)), k
    assert sorted(loss_a) == sorted(loss_b)
    for k in loss_a:
        assert abs(loss_a[k] - loss_b[k]) <= 2e-6 * abs(loss_b[k]) + 1e-12, (k, loss_a[k], loss_b[k])
    scale = float(grad_b.abs().max())
    assert float((grad_a - grad_b).abs().max()) <= 1e-5 * scale  # atomically accumulated sums (table scatter, weight gradients) in another order
    assert torch.equal(grad_a == 0, grad_b == 0)  # the same parameters receive a gradient


@pytest.mark.parametrize("size", ["tiny", "default", "default256"])  # "default": the reference at 16 x 2^19 / 5 x 2^17 tables, 64 rays
@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_density_on_identical_samples(golden_dir, mode, size):
    """The strict bound: on the reference's OWN sample bins (stored in the golden file) density must agree to 1e-4 and RGB(T) to 1e-3."""
    from nerfstudio_thermal_amd import ops

    g = golden_file(golden_dir, mode, size)
    _, _, _, eng = build(mode, size)
    gi, o, d, cam = dev_inputs(golden_dir, size)
    for s, fld, pose, frozen in (("", eng.field, eng.pose, eng.frozen_rgb), ("_thermal", eng.field_thermal, eng.pose_thermal, eng.frozen_thermal)):
        if fld is None:
            continue
        po, pd = ops.pose_apply_fwd(pose, frozen, cam, o, d)
        e2 = torch.from_numpy(g[f"train/ebins{s}_2"]).to(DEV).contiguous()
        dens, rgb, _ = ops.field_fwd(fld, po, pd, cam, e2, True)
        assert md(dens, g[f"train/density{s}"][..., 0]) <= DENS_TOL, (s, md(dens, g[f"train/density{s}"][..., 0]))
        w = torch.from_numpy(g[f"train/weights{s}_2"]).to(DEV).contiguous()
        comp = ops.composite_fwd(rgb, w, e2, True, want_depth=False)[0]
        assert md(comp, g[f"train/rgb{s}"] if mode == "separate" else g["train/rgbt"]) <= RGB_TOL


def test_full_size_properties():
    """BASELINE config sizes (4096 rays, 256/96/48 samples, 2^19 / 2^17 tables): size-independent properties."""
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig

    cfg = ThermalNerfactoModelConfig(density_mode="shared")
    arena = ParamArena(cfg, 8, DEV)
    shapes = {n: s for n, (_, s) in arena.layout.items()}
    arena.load({k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, seed=0).items()})
    eng = RenderEngine(cfg, arena, 8, [0, 0, 0, 0, 1, 1, 1, 1])
    N = 4096
    r = synth.synth_rays_simple(N)
    o, d, cam = (torch.from_numpy(r[k]).to(DEV) for k in ("origins", "directions", "camera_indices"))
    out, br = eng.get_outputs(o, d, cam, training=False)
    lv = br[""].levels
    for L in lv:
        assert bool((L.s_bins[:, 1:] >= L.s_bins[:, :-1]).all())  # sortedness of every level's bins
        assert float(L.s_bins.min()) >= 0.0 and float(L.s_bins.max()) <= 1.0
        assert bool(torch.isfinite(L.weights).all()) and float(L.weights.min()) >= 0.0
        assert float(L.weights.sum(-1).max()) <= 1.0 + 1e-4  # weights are a sub-probability distribution along the ray
    assert float(out["rgbt"].min()) >= 0.0 and float(out["rgbt"].max()) <= 1.0
    # determinism / idempotence of the eval path
    out2, _ = eng.get_outputs(o, d, cam, training=False)
    assert torch.equal(out["rgbt"], out2["rgbt"]) and torch.equal(out["density"], out2["density"])
    # ray independence: rendering a permuted batch permutes the result (expected depth excluded: it clips to batch-global range)
    perm = torch.randperm(N, device=DEV)
    out3, _ = eng.get_outputs(o[perm].contiguous(), d[perm].contiguous(), cam[perm].contiguous(), training=False)
    assert md(out3["rgbt"], out["rgbt"][perm]) == 0.0
    assert md(out3["density"], out["density"][perm]) == 0.0
    # chunk independence of everything but expected_depth
    half, _ = eng.get_outputs(o[: N // 2].contiguous(), d[: N // 2].contiguous(), cam[: N // 2].contiguous(), training=False)
    assert md(half["rgbt"], out["rgbt"][: N // 2]) == 0.0


@pytest.mark.parametrize("counts", [((128, 64), 32), ((200, 80), 50)])  # (52 rays x 50 samples = 81.25 of the field's 32-sample tiles)
def test_non_default_config_matches_oracle(golden_dir, counts):
    """Everything the goldens pin is at the reference's default hyper-parameters; this runs a differently configured model (sample counts,
    planes, resolutions, loss multipliers) against the live oracle: eval render, train-mode forward, every loss term, one fused step."""
    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig

    ocfg = tiny_cfg("shared", num_proposal_samples_per_ray=counts[0], num_nerf_samples_per_ray=counts[1], near_plane=0.1, far_plane=100.0,
                    max_res=1024, prop_max_res=(64, 128), distortion_loss_mult=0.01, interlevel_loss_mult=0.5, thermal_loss_mult=10.0,
                    tv_pixel_loss_mult=1e-3, cross_channel_loss_mult=1e-3)
    cfg = ThermalNerfactoModelConfig(density_mode="shared", log2_hashmap_size=ocfg.log2_hashmap_size, max_res=1024,
                                     num_proposal_samples_per_ray=counts[0], num_nerf_samples_per_ray=counts[1], near_plane=0.1, far_plane=100.0,
                                     distortion_loss_mult=0.01, interlevel_loss_mult=0.5, thermal_loss_mult=10.0, tv_pixel_loss_mult=1e-3,
                                     cross_channel_loss_mult=1e-3)
    for a, mr in zip(cfg.proposal_net_args_list, (64, 128)):
        a["log2_hashmap_size"] = ocfg.prop_log2_hashmap_size
        a["max_res"] = mr
    params = make_params(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load(params)
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    N = 52
    cams = synth.synth_cameras()
    idx = torch.from_numpy(synth.synth_ray_indices(cams, 64, seed=5))[:N].contiguous()  # 52 rays = 13 patches: not a multiple of 16
    t = lambda k: torch.from_numpy(cams[k])  # noqa: E731
    ro, rd, _, _ = orc.generate_rays(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
    cam = idx[:, 0].contiguous()
    dev = lambda x: x.to(DEV).contiguous()  # noqa: E731
    with torch.no_grad():
        ref = orc.get_outputs(params, ocfg, ro, rd, cam, training=False)
    out, _ = eng.get_outputs(dev(ro), dev(rd), dev(cam), training=False)
    for key in ("rgb", "rgb_thermal", "accumulation"):
        assert float((out[key].cpu() - ref[key]).abs().max()) <= RGB_TOL, key
    # (no golden for this configuration: the default configuration's measured 1-ulp response stands in for its own)
    s1, su = chain_sens(golden_dir, "shared", "tiny")
    assert_density_chain(out["density"], ref["density"], "density", s1[""], su[""])
    # training forward + losses with injected jitters
    jit = [torch.from_numpy(j) for j in synth.synth_jitters(N, seed=77)]
    img, is_th = (torch.from_numpy(a) for a in synth.synth_gt(idx.numpy(), cams, seed=3))
    p_req = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref_t = orc.get_outputs(p_req, ocfg, ro, rd, cam, training=True, anneal=1.0, jitters=jit)
    ref_l = orc.loss_dict(p_req, ocfg, ref_t, img, is_th)
    eng.anneal = 1.0
    arena.zero_grad()
    out_t, branches = eng.get_outputs(dev(ro), dev(rd), dev(cam), True, [dev(j.reshape(-1)) for j in jit])
    losses = eng.loss_and_backward(out_t, branches, dev(cam), dev(img), dev(is_th))
    assert sorted(losses) == sorted(ref_l)
    for k in ref_l:
        a, b = float(losses[k]), float(ref_l[k])
        assert abs(a - b) <= 5e-4 * abs(b) + 1e-9, (k, a, b)
    sum(ref_l.values()).backward()
    # gradient norms per parameter (chained gradients: robust statistic, see test_train_step_matches_reference_golden)
    for name in arena.names():
        g_ref = p_req[name].grad
        got = arena.grad_view(name)
        if g_ref is None:
            assert float(got.abs().max()) == 0.0, name
            continue
        rn = float(g_ref.double().norm())
        assert abs(float(got.double().norm()) - rn) <= 2e-2 * rn + 1e-12, (name, float(got.double().norm()), rn)
    # the pose corrections' gradient entry by entry: every ray's d origins / d directions lands in its camera's row (a partial last tile of the
    # field's d position pass once dropped its share of the last ray)
    name = next(n for n in arena.names() if n.endswith("pose_adjustment"))
    g_ref, got = p_req[name].grad, arena.grad_view(name).cpu()
    assert float((got - g_ref).abs().max()) <= 2e-2 * float(g_ref.abs().max()), (got, g_ref)


@pytest.mark.parametrize("rays", [4096, 8192])  # 8192 = the batch BASELINE config 2 names
def test_full_size_separate_mode_run_is_stable(rays):
    """BASELINE config 2 (separate density fields + density loss, default table sizes): 80 fused steps stay finite and the losses fall."""
    import bench

    dev = torch.device(DEV, 0)
    torch.manual_seed(4321)
    cfg, arena, eng = bench.build_engine(dev, mode="separate")
    cam_t, _, _, _ = bench.make_batch(dev, rays, 42)
    cache = bench.make_image_cache(dev)
    hist = []
    for step in range(80):
        losses = bench.one_step(eng, cam_t, cache, rays, step, None)
        if step % 20 == 0 or step == 79:
            hist.append({k: float(v) for k, v in losses.items()})
    assert all(np.isfinite(v) for h in hist for v in h.values()), hist[-1]
    assert "density_loss" in hist[-1] and "camera_opt_regularizer_thermal" in hist[-1]
    assert hist[-1]["rgb_loss"] < hist[0]["rgb_loss"] / 2 and hist[-1]["thermal_loss"] < hist[0]["thermal_loss"] / 2, (hist[0], hist[-1])
    assert eng.group_steps["fields_thermal"] == 80 and eng.group_steps["proposal_networks_thermal"] == 80  # the thermal sampler always updates
    assert bool(torch.isfinite(arena.params).all())


def test_full_size_training_run_is_stable():
    """BASELINE config 1 end to end (default table sizes, 4096 rays, new pixels every step as in bench.py): 300 fused steps stay finite,
    the photometric losses fall, the proposal networks are stepped exactly on the scheduled iterations, and the result is reproducible
    up to float-atomic noise."""
    import bench

    dev = torch.device(DEV, 0)
    finals = []
    for rep in range(2):
        torch.manual_seed(1234)
        cfg, arena, eng = bench.build_engine(dev)
        cam_t, _, _, _ = bench.make_batch(dev, 4096, 42)
        cache = bench.make_image_cache(dev)
        hist = []
        for step in range(300):
            losses = bench.one_step(eng, cam_t, cache, 4096, step, None)
            if step % 20 == 0 or step == 299:
                hist.append({k: float(v) for k, v in losses.items()})
        assert all(np.isfinite(v) for h in hist for v in h.values()), hist[-1]
        first, last = hist[0], hist[-1]
        assert last["rgb_loss"] < first["rgb_loss"] / 3 and last["thermal_loss"] < first["thermal_loss"] / 3, (first, last)
        assert eng.group_steps["fields"] == 300 and 10 < eng.group_steps["proposal_networks"] < 300  # proposal nets: only on scheduled steps
        assert bool(torch.isfinite(arena.params).all())
        finals.append(last)
    # two runs differ only by float-atomic ordering, which training amplifies (chaotic): the photometric losses agree to a few percent, the tiny
    # regularisers (1e-5) wander by tens of percent and are only required to be finite (above)
    for k in ("rgb_loss", "thermal_loss"):
        assert abs(finals[0][k] - finals[1][k]) <= 0.25 * abs(finals[0][k]) + 1e-8, (k, finals[0][k], finals[1][k])


def init_scale_params(ocfg, seed=5):
    """Parameters at the scale nerfstudio initialises them with: hash tables U(-1e-4, 1e-4) (field_components/encodings.py:355-357),
    torch.nn.Linear's default U(-1/sqrt(in), 1/sqrt(in)) for weights and biases, N(0, 1) embeddings, zero pose adjustments."""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in orc.param_shapes(ocfg).items():
        if name.endswith("hash_table"):
            v = g.uniform(-1e-4, 1e-4, shape)
        elif "pose_adjustment" in name:
            v = np.zeros(shape)
        elif "embedding" in name:
            v = g.standard_normal(shape)
        else:  # Linear weight [out, in] or its bias [out]: both U(+-1/sqrt(fan_in))
            wshape = shape if len(shape) == 2 else orc.param_shapes(ocfg)[name[: -len("bias")] + "weight"]
            v = g.uniform(-1.0, 1.0, shape) / np.sqrt(wshape[1])
        out[name] = torch.from_numpy(v.astype(np.float32))
    return out


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_density_chain_strict_on_init_scale_tables(golden_dir, mode):
    """north_star's bound, unrelaxed and through the WHOLE chain (pose -> two PDF resamplings -> field -> compositing): density 1e-4 on EVERY
    sample, RGB/thermal 1e-3, on identical rays, with parameters at the scale training starts from.  (The relaxed chain bound of
    assert_density_chain is specific to the deliberately high-variance synthetic tables of the golden sets, where one fp32 ulp in a sample
    position moves the reference's own density by more than 1e-4.)"""
    ocfg = size_cfg("tiny", mode)
    cfg = pkg_cfg(ocfg)
    params = init_scale_params(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load(params)
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    gi, o, d, cam = dev_inputs(golden_dir, "tiny")
    with torch.no_grad():
        ref = orc.get_outputs(params, ocfg, gi["origins"], gi["directions"], gi["camera_indices"], training=False)
    out, _ = eng.get_outputs(o, d, cam, training=False)
    for s in ("", "_thermal") if mode == "separate" else ("",):
        err = (out[f"density{s}"].cpu() - ref[f"density{s}"]).abs()
        assert float(err.max()) <= DENS_TOL, (s, float(err.max()))
        assert float(ref[f"density{s}"].max()) > 1e-2  # the densities are not trivially zero
    assert md(out["rgb"], ref["rgb"]) <= RGB_TOL and md(out["rgb_thermal"], ref["rgb_thermal"]) <= RGB_TOL
    assert md(out["expected_depth"], ref["expected_depth"]) <= 1e-3


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_density_chain_strict_default_table_size(golden_dir, mode):
    """The unrelaxed whole-chain bound (density 1e-4 on EVERY sample, RGB / thermal 1e-3) at the DEFAULT table sizes (16 x 2^19, 5 x 2^17) with
    parameters at the scale training starts from, 256 rays: the configuration ns-train ships."""
    ocfg = size_cfg("default256", mode)
    cfg = pkg_cfg(ocfg)
    params = init_scale_params(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load(params)
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    gi, o, d, cam = dev_inputs(golden_dir, "default256")
    with torch.no_grad():
        ref = orc.get_outputs(params, ocfg, gi["origins"], gi["directions"], gi["camera_indices"], training=False)
    out, _ = eng.get_outputs(o, d, cam, training=False)
    for s in ("", "_thermal") if mode == "separate" else ("",):
        err = (out[f"density{s}"].cpu() - ref[f"density{s}"]).abs()
        assert float(err.max()) <= DENS_TOL, (s, float(err.max()))
        assert float(ref[f"density{s}"].max()) > 1e-2
    assert md(out["rgb"], ref["rgb"]) <= RGB_TOL and md(out["rgb_thermal"], ref["rgb_thermal"]) <= RGB_TOL
    assert md(out["expected_depth"], ref["expected_depth"]) <= 1e-3


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_density_after_training(golden_dir, mode):
    """Weights that 200 training iterations produced (tiny tables, init-scale start), read back from the arena and given to the oracle (with the
    sampler's annealing exponent of that iteration).  (a) On IDENTICAL sample positions -- the engine's own final bins fed to the oracle's
    field -- every sample's density holds north_star's bound unrelaxed: 1e-4 absolute below 1, 1e-4 relative above (a trained logit makes
    exp() large: thousands in separate mode).  (b) Through the WHOLE chain RGB / thermal hold 1e-3 on every ray (measured ~1e-5); the chained
    density is measured against what the chain itself defines: the share of samples beyond 1e-4 (relative to scale) at most 3 % or 3x the
    share the ORACLE itself moves that far when its ray origins or directions move by one ulp, and none further off than 10x the ORACLE's own largest response to such a 1-ulp move (or 1e-3) -- trained tables are high-variance, a resampled bin that
    moves by one ulp moves the reference's own density by up to 1e-2 there (scripts/diag_trained_chain.py)."""
    from nerfstudio_thermal_amd import ops

    ocfg = size_cfg("tiny", mode)
    cfg = pkg_cfg(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load(init_scale_params(ocfg))
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    gi, o, d, cam = dev_inputs(golden_dir, "tiny")
    img, is_th = gi["image"].to(DEV), gi["is_thermal"].to(DEV)
    first = last = None
    for step in range(200):
        losses = eng.train_step(o, d, cam, img, is_th, step)
        tot = float(sum(float(v) for v in losses.values()))
        first = tot if first is None else first
        last = tot
    assert np.isfinite(last) and last < first  # it trained
    params = {k: arena.view(k).detach().cpu().clone() for k in arena.names()}
    with torch.no_grad():
        ref = orc.get_outputs(params, ocfg, gi["origins"], gi["directions"], gi["camera_indices"], training=False, anneal=eng.anneal)
        # the oracle against ITSELF with the ray origins / directions moved by one fp32 ulp either way: how far the chain's own arithmetic
        # defines its densities (four probes: the trained weights differ from run to run -- float atomics in the 200 iterations -- and one
        # probe alone caught the chain's sharpest response in some runs and missed it in others)
        hi, lo = torch.tensor(9.0), torch.tensor(-9.0)
        ref_ulps = [orc.get_outputs(params, ocfg, oo, dd_, gi["camera_indices"], training=False, anneal=eng.anneal)
                    for oo, dd_ in ((torch.nextafter(gi["origins"], hi), gi["directions"]), (torch.nextafter(gi["origins"], lo), gi["directions"]),
                                    (gi["origins"], torch.nextafter(gi["directions"], hi)), (gi["origins"], torch.nextafter(gi["directions"], lo)))]
    out, branches = eng.get_outputs(o, d, cam, training=False)
    assert md(out["rgb"], ref["rgb"]) <= RGB_TOL and md(out["rgb_thermal"], ref["rgb_thermal"]) <= RGB_TOL
    for s, prefix in (("", "field"), ("_thermal", "field_thermal")) if mode == "separate" else (("", "field"),):
        br = branches[s]
        e2 = br.levels[2].e_bins
        # (a) identical samples: the oracle's field on the engine's bins and pose-corrected rays
        with torch.no_grad():
            pos = orc.Samples(s_bins=e2.cpu(), e_bins=e2.cpu()).positions(br.origins.cpu(), br.directions.cpu())
            dref, _, _, _ = orc.field_density(params, prefix, ocfg, pos)
        err = (br.levels[2].density.cpu() - dref[..., 0]).abs()
        tol = DENS_TOL * torch.clamp(dref[..., 0].abs(), min=1.0)
        assert bool((err <= tol).all()), (s, "identical samples", float((err / tol).max()))
        assert float(dref.max()) > 0.1  # training produced real densities
        # (b) whole chain, every sample
        scale = torch.clamp(ref[f"density{s}"].abs(), min=1.0)
        cerr = (out[f"density{s}"].cpu() - ref[f"density{s}"]).abs() / scale
        serrs = [(r[f"density{s}"] - ref[f"density{s}"]).abs() / scale for r in ref_ulps]
        sens, sens_frac = max(float(e.max()) for e in serrs), max(float((e > DENS_TOL).float().mean()) for e in serrs)
        frac = float((cerr > DENS_TOL).float().mean())
        assert frac <= max(0.03, 3.0 * sens_frac), (s, "chain", frac, sens_frac)
        assert float(cerr.max()) <= max(1e-3, 10.0 * sens), (s, "chain", float(cerr.max()), sens)


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_render_rays_train_bwd_is_the_launch_sequence(golden_dir, mode, monkeypatch):
    """tn_render_rays_train_bwd (one call of the C ABI per branch: renderer backward, field backward with d position and table scatter, both
    proposal networks on the library's companion streams) against the same backward made of the individual entry points on torch side streams
    (TN_ONE_CALL_BWD=0): every loss and every gradient, on an iteration that updates the proposal networks and on one that does not.  Identical
    up to the order of the float atomics (weight-gradient block sums, d origins / d directions)."""
    from nerfstudio_thermal_amd import engine as E

    gi, o, d, cam = dev_inputs(golden_dir, "tiny")
    img, is_th = gi["image"].to(DEV), gi["is_thermal"].to(DEV)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]

    def grads(one_call, prop_update):
        monkeypatch.setattr(E, "_ONE_CALL_BWD", one_call)
        _, _, arena, eng = build(mode, "tiny")
        if not prop_update:  # make the sampler skip the proposal networks' gradient this iteration (ray_samplers.py:591)
            eng.sampler_step, eng.steps_since_update = 2000, 0
        arena.zero_grad()
        out, br = eng.get_outputs(o, d, cam, True, jit, jit_t)
        assert br[""].prop_grad == prop_update
        losses = eng.loss_and_backward(out, br, cam, img, is_th)
        torch.cuda.synchronize()
        return {k: float(v) for k, v in losses.items()}, arena.grads.clone()

    for prop_update in (True, False):
        la, ga = grads(True, prop_update)
        lb, gb = grads(False, prop_update)
        assert la.keys() == lb.keys()
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-6 * abs(lb[k]) + 1e-12, (k, la[k], lb[k])
        assert torch.equal(ga == 0, gb == 0)  # same zero pattern
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()), float((ga - gb).abs().max())


def _ulp_distance(a, b):
    """distance in fp32 ulps between two positive float tensors (their bit patterns are ordered like the values)"""
    ai = torch.as_tensor(a).float().contiguous().view(torch.int32).long()
    bi = torch.as_tensor(b).float().contiguous().view(torch.int32).long()
    return (ai - bi).abs()


@pytest.mark.parametrize("size", ["tiny", "default256"])
@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_sampler_bins_ulp_distance(golden_dir, mode, size, capsys):
    """How far the HIP sampler's bins sit from the REFERENCE's (train forward with the golden's injected jitter), in fp32 ulps of the euclidean
    bins, level by level: level 0 must be exact up to one ulp (closed-form spacing), the two PDF resamplings add a few ulps each, and a handful
    of bins per thousand land in a neighbouring CDF interval (searchsorted on a value within an ulp of a knot: the 'outliers' the bin tests
    count).  The histogram is printed (pytest -s) and quoted in DESIGN.md section 4; the chained density bound K_CHAIN x S rests on it."""
    g = golden_file(golden_dir, mode, size)
    _, _, _, eng = build(mode, size)
    gi, o, d, cam = dev_inputs(golden_dir, size)
    jit = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters"]]
    jit_t = [j.to(DEV).reshape(-1).contiguous() for j in gi["jitters_thermal"]]
    eng.set_anneal_for_step(500)
    _, branches = eng.get_outputs(o, d, cam, True, jit, jit_t)
    edges = [0, 1, 2, 4, 8, 64]
    checks = []
    for sfx in ("", "_thermal") if mode == "separate" else ("",):
        for i, lv in enumerate(branches[sfx].levels):
            ref = g[f"train/ebins{sfx}_{i}"] if f"train/ebins{sfx}_{i}" in g.files else None
            if ref is None:
                continue
            dist = _ulp_distance(lv.e_bins.detach().cpu(), ref).reshape(-1)
            n = dist.numel()
            hist = [int((dist == 0).sum())] + [int(((dist > lo) & (dist <= hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])] + [int((dist > edges[-1]).sum())]
            with capsys.disabled():
                print(f"\n[ulp distance of e_bins, {mode} {size} branch '{sfx}' level {i}] 0: {hist[0]}  1: {hist[1]}  2: {hist[2]}  3-4: {hist[3]}  5-8: {hist[4]}  "
                      f"9-64: {hist[5]}  >64: {hist[6]}  of {n}")
            checks.append((sfx, i, dist, hist))
    for sfx, i, dist, hist in checks:
        if i == 0:
            assert int(dist.max()) <= 2, (sfx, i, int(dist.max()))
        else:
            assert float((dist <= 8).double().mean()) >= 0.90, (sfx, i, hist)
            assert float((dist > 64).double().mean()) <= 0.02, (sfx, i, hist)
