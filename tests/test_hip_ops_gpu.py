"""GPU parity tests, op by op: every C-ABI entry point against the CPU oracle on identical seeded inputs.

Tolerances (BASELINE.json north_star): RGB/thermal 1e-3 abs, density 1e-4 abs.  The per-op checks here are tighter wherever the
arithmetic allows (most ops agree to a few fp32 ulps); discontinuous ops (searchsorted in the PDF sampler, median depth) report
an outlier fraction instead of a max-abs.
"""
import numpy as np
import pytest
import torch

import thermal_nerfacto_oracle as orc
from helpers import make_params, tiny_cfg, SEED
from nerfstudio_thermal_amd import ops, synth
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
from nerfstudio_thermal_amd.netparams import field_params, prop_params

pytestmark = pytest.mark.gpu
DEV = "cuda"


def pkg_cfg(ocfg: orc.OracleConfig) -> ThermalNerfactoModelConfig:
    c = ThermalNerfactoModelConfig(density_mode=ocfg.density_mode, log2_hashmap_size=ocfg.log2_hashmap_size)
    for a in c.proposal_net_args_list:
        a["log2_hashmap_size"] = ocfg.prop_log2_hashmap_size
    return c


def setup_pair(mode="shared", **kw):
    ocfg = tiny_cfg(mode, **kw)
    params = make_params(ocfg)
    cfg = pkg_cfg(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load(params)
    return ocfg, params, cfg, arena


def rays(n, seed=11):
    r = synth.synth_rays_simple(n, seed)
    return {k: torch.from_numpy(v) for k, v in r.items()}


def g(x):
    return x.to(DEV).contiguous()


def md(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs().max())


def outlier_fraction(a, b, tol):
    d = (a.detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs()
    return float((d > tol).double().mean())


# ------------------------------------------------------------------------------------------------ a1 / a4
def test_raygen_matches_oracle():
    cams = synth.synth_cameras()
    idx = torch.from_numpy(synth.synth_ray_indices(cams, 256))
    t = lambda k: torch.from_numpy(cams[k])  # noqa: E731
    ro, rd, ra, rn = orc.generate_rays(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
    o, d, a, n = ops.raygen(g(idx), g(t("c2w")), g(t("fx")), g(t("fy")), g(t("cx")), g(t("cy")), g(t("distortion")))
    assert md(o, ro) == 0.0
    assert md(d, rd) < 2e-6
    assert md(a, ra) < 1e-9
    assert md(n, rn) < 1e-5
    # no distortion
    ro, rd, ra, rn = orc.generate_rays(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), None)
    o, d, a, n = ops.raygen(g(idx), g(t("c2w")), g(t("fx")), g(t("fy")), g(t("cx")), g(t("cy")), None)
    assert md(d, rd) < 1e-6


def test_pose_apply_fwd_bwd():
    N, C = 512, 8
    r = rays(N)
    pose = torch.from_numpy(synth.uniform("pose_t", (C, 6), -0.2, 0.2, 5))
    pose[1, 3:] = 0.0
    pose[2, 3:] = torch.tensor([1e-3, -2e-3, 5e-4])
    pose.requires_grad_(True)
    frozen = torch.tensor([0, 0, 0, 0, 1, 1, 1, 1], dtype=torch.bool)
    ro, rd = orc.apply_pose_adjustment(pose, frozen, r["camera_indices"], r["origins"], r["directions"])
    go = torch.from_numpy(synth.uniform("go", (N, 3), seed=5))
    gd = torch.from_numpy(synth.uniform("gd", (N, 3), seed=5))
    ((ro * go).sum() + (rd * gd).sum()).backward()
    o, d = ops.pose_apply_fwd(g(pose.detach()), g(frozen.to(torch.uint8)), g(r["camera_indices"]), g(r["origins"]), g(r["directions"]))
    assert md(o, ro) < 1e-6 and md(d, rd) < 1e-6
    gp = torch.zeros((C, 6), device=DEV)
    ops.pose_apply_bwd(g(pose.detach()), g(frozen.to(torch.uint8)), g(r["camera_indices"]), g(r["directions"]), g(go), g(gd), gp)
    assert md(gp, pose.grad) < 2e-4 * float(pose.grad.abs().max())


# ------------------------------------------------------------------------------------------------ samplers / weights
@pytest.mark.parametrize("train", [False, True])
def test_spaced_pdf_weights_chain(train):
    N = 300  # not a multiple of the 4 rays per workgroup
    j0, j1, j2 = (torch.from_numpy(j) for j in synth.synth_jitters(N))
    nears = torch.ones(N, 1) * (0.05 if train else 0.0)
    fars = torch.ones(N, 1) * 1000.0
    s0 = orc.spaced_bins(N, 256, j0 if train else None)
    e0 = orc.s_to_euclidean(s0, nears, fars)
    hs0, he0 = ops.spaced_bins(g(nears), g(fars), 256, g(j0) if train else None)
    assert md(hs0, s0) == 0.0
    assert md(he0, e0) <= 1e-6 * float(e0.max())
    dens = torch.from_numpy(synth.uniform("dens0", (N, 256, 1), 0.0, 60.0, SEED)) ** 2 / 60.0
    dens[5] = 0.0  # empty ray -> PDF padding guard
    smp = orc.Samples(s_bins=s0, e_bins=e0)
    w0 = orc.get_weights(smp.deltas, dens)
    hw0, med0 = ops.weights_fwd(g(e0), g(dens[..., 0]), want_median=True)
    assert md(hw0, w0[..., 0]) < 2e-6
    ref_med = orc.depth_median(w0, smp)
    assert outlier_fraction(med0, ref_med, 1e-6) <= 0.01
    for anneal in (1.0, 0.37):
        s1 = orc.pdf_resample(s0, torch.pow(w0, anneal), 96, j1 if train else None)
        hs1, he1 = ops.pdf_resample(g(s0), g(w0[..., 0]), 96, anneal, g(nears), g(fars), g(j1) if train else None)
        assert outlier_fraction(hs1, s1, 2e-6) <= 0.005, anneal
        e1 = orc.s_to_euclidean(s1, nears, fars)
        assert outlier_fraction(he1 / e1.to(DEV), torch.ones_like(e1), 1e-4) <= 0.005
    # second resample 96 -> 48 (ITEMS=2 path)
    smp1 = orc.Samples(s_bins=s1, e_bins=e1)
    dens1 = torch.from_numpy(synth.uniform("dens1", (N, 96, 1), 0.0, 30.0, SEED))
    w1 = orc.get_weights(smp1.deltas, dens1)
    hw1, _ = ops.weights_fwd(g(e1), g(dens1[..., 0]))
    assert md(hw1, w1[..., 0]) < 2e-6
    s2 = orc.pdf_resample(s1, w1, 48, j2 if train else None)
    hs2, _ = ops.pdf_resample(g(s1), g(w1[..., 0]), 48, 1.0, g(nears), g(fars), g(j2) if train else None)
    assert outlier_fraction(hs2, s2, 2e-6) <= 0.005
    # the fused entry point (weights of a level + bins of the next, one launch) is bit-identical to the two calls, for every ITEMS variant
    for Sp, Sn, hs, he, hd, jit, ann in ((256, 96, hs0, he0, g(dens[..., 0]), j1, 0.37), (96, 48, g(s1), g(e1), g(dens1[..., 0]), j2, 1.0),
                                         (96, 20, g(s1), g(e1), g(dens1[..., 0]), j2, 1.0)):
        jj = g(jit) if train else None
        w_a, med_a = ops.weights_fwd(he, hd, want_median=True)
        s_a, e_a = ops.pdf_resample(hs, w_a, Sn, ann, g(nears), g(fars), jj)
        w_b, med_b, s_b, e_b = ops.weights_resample(he, hd, hs, Sn, ann, g(nears), g(fars), jj)
        for a, b in ((w_a, w_b), (med_a, med_b), (s_a, s_b), (e_a, e_b)):
            assert torch.equal(a, b), (Sp, Sn)
    w64 = ops.weights_resample(he0[:, :65].contiguous(), g(dens[:, :64, 0]), hs0[:, :65].contiguous(), 32, 1.0, g(nears), g(fars), None, want_median=False)
    assert w64[1] is None and w64[2].shape == (N, 33)  # ITEMS = 1 path, median optional


@pytest.mark.parametrize("S", [48, 96, 256, 7])
def test_weights_bwd(S):
    N = 130
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s = orc.spaced_bins(N, S, None)
    e = orc.s_to_euclidean(s, nears, fars)
    dens = (torch.from_numpy(synth.uniform(f"wd{S}", (N, S, 1), 0.0, 8.0, SEED))).requires_grad_(True)
    smp = orc.Samples(s_bins=s, e_bins=e)
    w = orc.get_weights(smp.deltas, dens)
    gw = torch.from_numpy(synth.uniform(f"gw{S}", (N, S, 1), seed=SEED))
    (w * gw).sum().backward()
    hw, _ = ops.weights_fwd(g(e), g(dens.detach()[..., 0]))
    dd = ops.weights_bwd(g(e), g(dens.detach()[..., 0]), hw, g(gw[..., 0]))
    assert md(dd, dens.grad[..., 0]) <= 1e-5 * max(1.0, float(dens.grad.abs().max()))


# ------------------------------------------------------------------------------------------------ proposal nets
def sample_level(N, S, nears, fars):
    s = orc.spaced_bins(N, S, None)
    return s, orc.s_to_euclidean(s, nears, fars)


@pytest.mark.parametrize("lvl,S,N", [(0, 256, 128), (1, 96, 128), (0, 7, 37), (1, 3, 5)])
def test_prop_density_fwd_bwd(lvl, S, N):
    """(the last two cases: 259 and 15 points -- the tail of the backward kernel's 64-sample MFMA tiles, and fewer points than one 16-sample block)"""
    ocfg, params, cfg, arena = setup_pair()
    r = rays(N)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o = r["origins"].clone().requires_grad_(True)
    d = r["directions"].clone().requires_grad_(True)
    dens = orc.prop_density(p, "proposal_networks", lvl, ocfg, smp.positions(o, d))
    net = prop_params(arena, "proposal_networks", lvl, cfg, with_grads=True)
    hd = ops.prop_density_fwd(net, g(r["origins"]), g(r["directions"]), g(e))
    assert md(hd, dens[..., 0]) <= 1e-4, md(hd, dens[..., 0])
    assert md(hd / dens[..., 0].detach().to(DEV).clamp_min(1e-6), torch.ones(N, S)) <= 2e-5
    gd = torch.from_numpy(synth.uniform(f"gdens{lvl}", (N, S, 1), seed=SEED))
    (dens * gd).sum().backward()
    arena.zero_grad()
    d_o = torch.zeros((N, 3), device=DEV)
    d_d = torch.zeros((N, 3), device=DEV)
    ops.prop_density_bwd(net, g(r["origins"]), g(r["directions"]), g(e), g(gd[..., 0]), d_o, d_d)
    k = orc.prop_keys("proposal_networks", lvl)
    for short in ("table", "w0", "b0", "w1", "b1"):
        ref = p[k[short]].grad
        got = arena.grad_view(k[short])
        scale = float(ref.abs().max())
        assert md(got, ref) <= 2e-4 * scale, (short, md(got, ref), scale)
    assert md(d_o, o.grad) <= 2e-4 * float(o.grad.abs().max())
    assert md(d_d, d.grad) <= 2e-4 * float(d.grad.abs().max())


def test_hash_scatter_accumulates_at_least_as_accurately_as_fp32():
    """The fold pass sums a bucket's contributions in DOUBLE (LDS atomic adds) and rounds once.  On a coarse 5-level grid where every live slot
    receives hundreds of contributions, the result must be closer to the exact (float64) gradient than the reference's own fp32 accumulation
    (autograd of the fp32 oracle) is, and within 2e-6 of it relative to the largest entry."""
    L, log2T, maxr, S, N = 5, 13, 256, 256, 512
    r = rays(N)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    table32 = torch.from_numpy(synth.uniform("hsd_table", (L * 2**log2T, 2), seed=SEED)) * 0.5
    res = orc.level_resolutions(L, 16, maxr)
    p, _ = orc.unit_cube_positions(smp.positions(r["origins"], r["directions"]))
    g_enc = torch.from_numpy(synth.uniform("hsd_g", (N * S, 2 * L), seed=SEED))
    grads = {}
    for dt in (torch.float32, torch.float64):  # same fp32 positions; the table, the interpolation weights and the sums in fp32 / in double
        t = table32.detach().clone().to(dt).requires_grad_(True)
        enc = orc.hash_encode(p.view(-1, 3).to(dt), t, res.to(dt), log2T)
        (enc * g_enc.to(dt)).sum().backward()
        grads[dt] = t.grad
    exact = grads[torch.float64]
    tg = torch.zeros((L * 2**log2T, 2), device=DEV)
    ops.hash_scatter(g(table32), tg, L, log2T, res.tolist(), g(r["origins"]), g(r["directions"]), g(e),
                     g(g_enc.reshape(N * S, L, 2).permute(1, 0, 2).contiguous()), None, None)
    scale = float(exact.abs().max())
    err_hip = float((tg.cpu().double() - exact).abs().max()) / scale
    err_ref = float((grads[torch.float32].double() - exact).abs().max()) / scale
    assert err_hip <= 2e-6, (err_hip, err_ref)
    assert err_hip <= err_ref, (err_hip, err_ref)
    assert torch.equal((tg == 0).cpu(), exact == 0)


# ------------------------------------------------------------------------------------------------ main field
@pytest.mark.parametrize("mode,training", [("shared", False), ("shared", True), ("separate", True)])
def test_field_fwd_bwd(mode, training):
    ocfg, params, cfg, arena = setup_pair(mode)
    N, S = 101, 48  # 4848 points = 151.5 of the 32-sample MFMA tiles: the last ray ends in a half-empty tile
    r = rays(N)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    prefix = "field_thermal" if mode == "separate" else "field"
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o = r["origins"].clone().requires_grad_(True)
    d = r["directions"].clone().requires_grad_(True)
    dens, geo, pre, enc = orc.field_density(p, prefix, ocfg, smp.positions(o, d))
    rgb = orc.field_color(p, prefix, ocfg, d.detach(), geo, r["camera_indices"], training)
    fld = field_params(arena, prefix, cfg, with_grads=True)
    hd, hrgb, hpre = ops.field_fwd(fld, g(r["origins"]), g(r["directions"]), g(r["camera_indices"]), g(e), training, want_pre=True)
    assert md(hpre, pre[..., 0]) <= 2e-5, md(hpre, pre[..., 0])
    assert md(hd, dens[..., 0]) <= 1e-4, md(hd, dens[..., 0])
    assert md(hrgb, rgb) <= 1e-4, md(hrgb, rgb)
    hd2 = ops.field_density_fwd(fld, g(r["origins"]), g(r["directions"]), g(e))
    assert md(hd2, hd) == 0.0
    if not training:
        return
    C = fld.num_channels
    gd = torch.from_numpy(synth.uniform("gfd", (N, S, 1), seed=SEED))
    gc = torch.from_numpy(synth.uniform("gfc", (N, S, C), seed=SEED))
    ((dens * gd).sum() + (rgb * gc).sum()).backward()
    arena.zero_grad()
    d_o = torch.zeros((N, 3), device=DEV)
    d_d = torch.zeros((N, 3), device=DEV)
    ops.field_bwd(fld, g(r["origins"]), g(r["directions"]), g(r["camera_indices"]), g(e), g(gd[..., 0]), g(gc), d_o, d_d)
    k = orc.field_keys(prefix)
    for short in ("table", "w0", "b0", "w1", "b1", "hw0", "hb0", "hw1", "hb1", "hw2", "hb2", "emb"):
        ref = p[k[short]].grad
        got = arena.grad_view(k[short])
        scale = float(ref.abs().max())
        assert md(got, ref) <= 3e-4 * scale, (short, md(got, ref), scale)
    assert md(d_o, o.grad) <= 3e-4 * float(o.grad.abs().max())
    assert md(d_d, d.grad) <= 3e-4 * float(d.grad.abs().max())


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_field_density_only_fwd_bwd(mode):
    """get_density alone with its own backward (density2 / density2_thermal of separate mode, models/thermal_nerfacto.py:447-458): every
    gradient of the density path vs oracle autograd, and nothing of the colour head is touched."""
    ocfg, params, cfg, arena = setup_pair(mode)
    N, S = 101, 48  # (151.5 tiles)
    r = rays(N)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    prefix = "field_thermal" if mode == "separate" else "field"
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o = r["origins"].clone().requires_grad_(True)
    d = r["directions"].clone().requires_grad_(True)
    dens, _, _, _ = orc.field_density(p, prefix, ocfg, smp.positions(o, d))
    gd = torch.from_numpy(synth.uniform("gfd_only", (N, S, 1), seed=SEED))
    (dens * gd).sum().backward()
    fld = field_params(arena, prefix, cfg, with_grads=True)
    hd = ops.field_density_fwd(fld, g(r["origins"]), g(r["directions"]), g(e), training=True, tag="cross")
    assert md(hd, dens[..., 0]) <= 1e-4
    arena.zero_grad()
    d_o = torch.zeros((N, 3), device=DEV)
    d_d = torch.zeros((N, 3), device=DEV)
    ops.field_bwd(fld, g(r["origins"]), g(r["directions"]), g(r["camera_indices"]), g(e), g(gd[..., 0]), None, d_o, d_d, tag="cross")
    k = orc.field_keys(prefix)
    for short in ("table", "w0", "b0", "w1", "b1"):
        ref = p[k[short]].grad
        scale = float(ref.abs().max())
        assert md(arena.grad_view(k[short]), ref) <= 3e-4 * scale, (short, md(arena.grad_view(k[short]), ref), scale)
    for short in ("hw0", "hb0", "hw1", "hb1", "hw2", "hb2", "emb"):
        assert p[k[short]].grad is None
        assert float(arena.grad_view(k[short]).abs().max()) == 0.0, short
    assert md(d_o, o.grad) <= 3e-4 * float(o.grad.abs().max())
    assert md(d_d, d.grad) <= 3e-4 * float(d.grad.abs().max())


def test_field_default_table_size_hash_parity():
    """Full 2^19-entry tables / 2047-resolution top level: exercises the uint32 hash against the oracle's int64 one."""
    ocfg = orc.OracleConfig(density_mode="shared")
    shapes = {k: v for k, v in orc.param_shapes(ocfg).items() if k.startswith("field.")}
    params = {k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, seed=SEED).items()}
    cfg = ThermalNerfactoModelConfig(density_mode="shared")
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load(params)
    N, S = 64, 48
    r = rays(N)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    with torch.no_grad():
        dens, geo, pre, _ = orc.field_density(params, "field", ocfg, smp.positions(r["origins"], r["directions"]))
        rgb = orc.field_color(params, "field", ocfg, r["directions"], geo, r["camera_indices"], False)
    fld = field_params(arena, "field", cfg)
    hd, hrgb, hpre = ops.field_fwd(fld, g(r["origins"]), g(r["directions"]), g(r["camera_indices"]), g(e), False, want_pre=True)
    assert md(hpre, pre[..., 0]) <= 2e-5
    assert md(hd, dens[..., 0]) <= 1e-4
    assert md(hrgb, rgb) <= 1e-4


def test_hash_scatter_matches_autograd_of_hash_encode():
    """tn_hash_scatter alone: d table and d position of the reference hash encoding, for a 5-level and a 16-level grid, N not a multiple of 4."""
    # replica plan of the scatter (tn_common.h): (5, 13) -> level 0 through dense replicas, levels 1-4 straight into the hashed gradient;
    # (16, 12) and (5, 10): tables smaller than level 0's 4913 cells -> no replicas at all; use_workspace=False -> plain atomics everywhere
    for L, log2T, maxr, S, N in ((5, 13, 256, 96, 37), (16, 12, 2048, 48, 101), (5, 10, 256, 96, 7)):
        r = rays(N)
        nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
        s, e = sample_level(N, S, nears, fars)
        smp = orc.Samples(s_bins=s, e_bins=e)
        table = (torch.from_numpy(synth.uniform(f"hs_table{L}", (L * 2**log2T, 2), seed=SEED)) * 0.5).requires_grad_(True)
        o = r["origins"].clone().requires_grad_(True)
        d = r["directions"].clone().requires_grad_(True)
        res = orc.level_resolutions(L, 16, maxr)
        p, sel = orc.unit_cube_positions(smp.positions(o, d))
        enc = orc.hash_encode(p.view(-1, 3), table, res, log2T)
        ld = 16 if L == 5 else 32
        g_enc = torch.zeros((N * S, ld))
        g_enc[:, : 2 * L] = torch.from_numpy(synth.uniform(f"hs_g{L}", (N * S, 2 * L), seed=SEED))
        (enc * g_enc[:, : 2 * L]).sum().backward()
        # with the workspace the coarse levels go through the dense replicas + k_dense_reduce, without it straight into the hashed gradient
        zero_patterns = []
        for use_ws in (True, False):
            tg = torch.zeros((L * 2**log2T, 2), device=DEV)
            d_o, d_d = torch.zeros((N, 3), device=DEV), torch.zeros((N, 3), device=DEV)
            ops.hash_scatter(g(table.detach()), tg, L, log2T, res.tolist(), g(r["origins"]), g(r["directions"]), g(e), g(g_enc), d_o, d_d,
                             use_workspace=use_ws)
            assert md(tg, table.grad) <= 2e-5 * float(table.grad.abs().max()), (L, use_ws)
            assert md(d_o, o.grad) <= 2e-4 * float(o.grad.abs().max()), (L, use_ws)
            assert md(d_d, d.grad) <= 2e-4 * float(d.grad.abs().max()), (L, use_ws)
            zero_patterns.append((tg == 0).cpu())
        # untouched entries must stay exactly zero on both paths (Adam's eps = 1e-15 turns any residue into a full-size update)
        assert torch.equal(zero_patterns[0], zero_patterns[1])
        assert torch.equal(zero_patterns[0], table.grad == 0)
        # the same gradient handed over level-major [L, P, 2] (ld = -1: what the main field's backward produces): identical result
        g_lm = g_enc[:, : 2 * L].reshape(N * S, L, 2).permute(1, 0, 2).contiguous()
        for use_ws in (True, False):
            tg2 = torch.zeros((L * 2**log2T, 2), device=DEV)
            d_o2, d_d2 = torch.zeros((N, 3), device=DEV), torch.zeros((N, 3), device=DEV)
            ops.hash_scatter(g(table.detach()), tg2, L, log2T, res.tolist(), g(r["origins"]), g(r["directions"]), g(e), g(g_lm), d_o2, d_d2, use_workspace=use_ws)
            assert md(tg2, table.grad) <= 2e-5 * float(table.grad.abs().max()), (L, use_ws, "level-major")
            assert md(d_o2, o.grad) <= 2e-4 * float(o.grad.abs().max()), (L, use_ws, "level-major")
            assert torch.equal((tg2 == 0).cpu(), zero_patterns[0])


def test_render_psnr_vs_oracle():
    """'render PSNR vs ref' half of the BASELINE metric: eval-mode render of 512 rays, HIP vs oracle on identical rays."""
    from nerfstudio_thermal_amd.engine import RenderEngine

    ocfg, params, cfg, arena = setup_pair("shared")
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    cams = synth.synth_cameras()
    idx = torch.from_numpy(synth.synth_ray_indices(cams, 512))
    t = lambda k: torch.from_numpy(cams[k])  # noqa: E731
    ro, rd, _, _ = orc.generate_rays(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
    with torch.no_grad():
        ref = orc.get_outputs(params, ocfg, ro, rd, idx[:, 0], training=False)
    out, _ = eng.get_outputs(g(ro), g(rd), g(idx[:, 0]), training=False)
    for key, refv in (("rgb", ref["rgb"]), ("rgb_thermal", ref["rgb_thermal"])):
        mse = float(((out[key].cpu().double() - refv.double()) ** 2).mean())
        psnr = 99.0 if mse == 0 else -10.0 * np.log10(mse)
        assert psnr >= 80.0, (key, psnr)  # 1e-3 abs everywhere would be 60 dB; measured > 100 dB
        assert md(out[key], refv) <= 1e-3


@pytest.mark.parametrize("mode,N", [("shared", 512), ("separate", 130)])
def test_render_rays_eval_is_the_launch_sequence(mode, N):
    """tn_render_rays_eval (one library call: bins -> proposal nets -> resampling -> field -> weights -> renderers) gives bit for bit what the
    same entry points give when the engine calls them one by one, for every output key of the inference render, in both density modes."""
    from nerfstudio_thermal_amd.engine import RenderEngine

    ocfg, params, cfg, arena = setup_pair(mode)
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    r = rays(N)
    o, d = g(r["origins"]), g(r["directions"])
    cam = g(torch.arange(N, dtype=torch.int64) % ocfg.num_images)
    nears, fars = eng._nears_fars(N, False)
    for props, fld, anneal in ((eng.props, eng.field, 0.37),) + (((eng.props_thermal, eng.field_thermal, 1.0),) if mode == "separate" else ()):
        a = eng.render_branch(props, fld, None, None, o, d, cam, nears, fars, False, anneal, None, prop_grad=False)
        b = eng.render_branch_eval(props, fld, o, d, cam, nears, fars, anneal)
        for name in ("comp", "accumulation", "depth", "expected_depth", "rgb_samples"):
            x, y = getattr(a, name), getattr(b, name)
            assert torch.equal(x.nan_to_num(nan=-7.0), y.nan_to_num(nan=-7.0)), name
        for la, lb in zip(a.levels, b.levels):
            for name in ("s_bins", "e_bins", "density", "weights", "median"):
                assert torch.equal(getattr(la, name), getattr(lb, name)), name
    out, _ = eng.get_outputs(o, d, cam, training=False)  # the engine's inference path goes through the single call
    assert out["rgb"].shape == (N, 3) and out["density"].shape == (N, 48, 1) and torch.isfinite(out["expected_depth"]).all()


@pytest.mark.parametrize("with_pose", [True, False])
def test_render_rays_train_is_the_launch_sequence(with_pose, monkeypatch):
    """tn_render_rays_train (the training forward of a branch as one library call, all results in one buffer) against the same entry points
    called one by one by the engine: every tensor a loss or the backward reads is bit-identical, with and without pose correction."""
    from nerfstudio_thermal_amd import engine as engine_mod
    from nerfstudio_thermal_amd.engine import RenderEngine

    ocfg, params, cfg, arena = setup_pair("shared")
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    N = 260
    r = rays(N)
    o, d = g(r["origins"]), g(r["directions"])
    cam = g(torch.arange(N, dtype=torch.int64) % ocfg.num_images)
    nears, fars = eng._nears_fars(N, True)
    gen = torch.Generator().manual_seed(9)
    jit = [g(torch.rand(N, generator=gen)) for _ in range(3)]
    pose = eng.pose if with_pose else None
    if with_pose:
        eng.pose.copy_(g(torch.from_numpy(synth.uniform("pose_rt", tuple(eng.pose.shape), -0.02, 0.02, SEED))))
    monkeypatch.setattr(engine_mod, "_FUSE", True)
    a = eng.render_branch(eng.props, eng.field, pose, eng.frozen_rgb, o, d, cam, nears, fars, True, 0.37, jit, prop_grad=True)
    act_a = eng.field.workspace(N * 48, True, "main").clone()
    monkeypatch.setattr(engine_mod, "_FUSE", False)
    b = eng.render_branch(eng.props, eng.field, pose, eng.frozen_rgb, o, d, cam, nears, fars, True, 0.37, jit, prop_grad=True)
    act_b = eng.field.workspace(N * 48, True, "main")
    for name in ("origins", "directions", "rgb_samples", "comp", "accumulation", "depth", "expected_depth"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for la, lb in zip(a.levels, b.levels):
        for name in ("s_bins", "e_bins", "density", "weights", "median"):
            assert torch.equal(getattr(la, name), getattr(lb, name)), name
    nb = int(ops._lib.load().tn_field_workspace_bytes(N * 48, 1)) - int(ops._lib.load().tn_hash_scatter_workspace_bytes(N * 48, 16))
    assert torch.equal(act_a[:nb], act_b[:nb])  # the activations kept for the backward (everything before the scatter scratch)


# ------------------------------------------------------------------------------------------------ renderers and losses
@pytest.mark.parametrize("C,S,training", [(4, 48, False), (4, 48, True), (3, 48, True), (1, 96, False), (4, 256, True)])
def test_composite_fwd_bwd(C, S, training):
    N = 257
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    dens = torch.from_numpy(synth.uniform(f"cd{S}", (N, S, 1), 0.0, 6.0, SEED)) ** 2
    w = orc.get_weights(smp.deltas, dens).requires_grad_(True)
    rgb = torch.from_numpy(synth.uniform(f"crgb{C}{S}", (N, S, C), 0.0, 1.0, SEED))
    if not training:
        rgb[3, 5, 0] = float("nan")
        rgb[4, 1, 0] = 3.0
    rgb.requires_grad_(True)
    comp = orc.composite_rgb(rgb, w, training)
    hc, hacc, hmed, hexp = ops.composite_fwd(g(rgb.detach()), g(w.detach()[..., 0]), g(e), training)
    assert md(hc, comp) <= 2e-6
    assert md(hacc, orc.accumulation(w)) <= 2e-6
    assert md(hexp, orc.depth_expected(w, smp)) <= 1e-5 * float(e.max())
    assert outlier_fraction(hmed, orc.depth_median(w, smp), 1e-6) <= 0.01
    if not training:
        return
    gc = torch.from_numpy(synth.uniform(f"gcomp{C}", (N, C), seed=SEED))
    (comp * gc).sum().backward()
    dw = torch.zeros((N, S), device=DEV)
    drgb = ops.composite_bwd(g(rgb.detach()), g(w.detach()[..., 0]), g(gc), dw)
    assert md(drgb, rgb.grad) <= 1e-6
    assert md(dw, w.grad[..., 0]) <= 1e-5


def test_proposal_losses():
    N = 190
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    j0, j1, j2 = (torch.from_numpy(j) for j in synth.synth_jitters(N))
    lv = []
    s_prev, w_prev = None, None
    for i, S in enumerate((256, 96, 48)):
        s = orc.spaced_bins(N, S, j0) if i == 0 else orc.pdf_resample(s_prev, w_prev, S, (j1, j2)[i - 1])
        e = orc.s_to_euclidean(s, nears, fars)
        smp = orc.Samples(s_bins=s, e_bins=e)
        dens = torch.from_numpy(synth.uniform(f"pl{i}", (N, S, 1), 0.0, 5.0, SEED)) ** 3
        w = orc.get_weights(smp.deltas, dens).detach().requires_grad_(True)
        lv.append((smp, w))
        s_prev, w_prev = s, w.detach()
    wl, sl = [w for _, w in lv], [s for s, _ in lv]
    dist = orc.distortion_loss(wl, sl) * 0.002
    inter = orc.interlevel_loss(wl, sl)
    (dist + inter).backward()
    loss = torch.zeros(8, device=DEV)
    dw2 = torch.zeros((N, 48), device=DEV)
    ops.distortion_loss(g(sl[2].s_bins), g(wl[2].detach()[..., 0]), 0.002, loss[0:1], dw2)
    dw0 = torch.zeros((N, 256), device=DEV)
    dw1 = torch.zeros((N, 96), device=DEV)
    ops.interlevel_loss(g(sl[2].s_bins), g(wl[2].detach()[..., 0]), g(sl[0].s_bins), g(wl[0].detach()[..., 0]), 1.0, loss[1:2], dw0)
    ops.interlevel_loss(g(sl[2].s_bins), g(wl[2].detach()[..., 0]), g(sl[1].s_bins), g(wl[1].detach()[..., 0]), 1.0, loss[1:2], dw1)
    assert abs(float(loss[0]) - float(dist)) <= 1e-5 * float(dist)
    assert abs(float(loss[1]) - float(inter)) <= 1e-4 * float(inter) + 1e-9
    assert md(dw2, wl[2].grad[..., 0]) <= 1e-5 * float(wl[2].grad.abs().max())
    assert md(dw0, wl[0].grad[..., 0]) <= 1e-4 * float(wl[0].grad.abs().max())
    assert md(dw1, wl[1].grad[..., 0]) <= 1e-4 * float(wl[1].grad.abs().max())
    # the fused entry point (distortion + both interlevel terms in one launch) produces the same values and gradients, zero pattern included
    loss_f = torch.zeros(8, device=DEV)
    f2, f0, f1 = torch.zeros_like(dw2), torch.zeros_like(dw0), torch.zeros_like(dw1)
    fine_s, fine_w = g(sl[2].s_bins), g(wl[2].detach()[..., 0])
    ops.proposal_losses(fine_s, fine_w, [(g(sl[0].s_bins), g(wl[0].detach()[..., 0]), f0), (g(sl[1].s_bins), g(wl[1].detach()[..., 0]), f1)],
                        0.002, 1.0, loss_f[0:1], loss_f[1:2], f2)
    assert abs(float(loss_f[0]) - float(loss[0])) <= 1e-6 * abs(float(loss[0])) and abs(float(loss_f[1]) - float(loss[1])) <= 1e-5 * abs(float(loss[1]))
    for a, b in ((f2, dw2), (f0, dw0), (f1, dw1)):
        assert torch.equal(a, b)  # per-ray arithmetic is identical, only the launch differs
    # gradients are optional per level (proposal networks that are not updated this step)
    loss_h = torch.zeros(8, device=DEV)
    ops.proposal_losses(fine_s, fine_w, [(g(sl[0].s_bins), g(wl[0].detach()[..., 0]), None), (g(sl[1].s_bins), g(wl[1].detach()[..., 0]), None)],
                        0.002, 1.0, loss_h[0:1], loss_h[1:2], None)
    assert abs(float(loss_h[1]) - float(loss[1])) <= 1e-5 * abs(float(loss[1]))


@pytest.mark.parametrize("C,S,N,training", [(4, 48, 4096, True), (4, 48, 257, False), (3, 96, 130, True), (1, 256, 66, True), (4, 48, 70001, True)])
def test_render_fwd_bwd_is_the_separate_launches(C, S, N, training):
    """tn_render_fwd / tn_render_bwd (one launch each) against tn_weights_fwd + tn_minmax_init + tn_composite_fwd + tn_clip_depth and
    tn_composite_bwd + tn_weights_bwd: bit-identical, including the batch-global depth clip (per-block min / max pairs reduced by the
    follow-up launch), for batches below and above one round of blocks."""
    gen = torch.Generator().manual_seed(5 + S)
    nears = torch.full((N, 1), 0.05) + 0.3 * torch.rand((N, 1), generator=gen)
    fars = torch.full((N, 1), 50.0) + 900.0 * torch.rand((N, 1), generator=gen)
    s, e = sample_level(N, S, nears, fars)
    dens = (6.0 * torch.rand((N, S), generator=gen)) ** 2
    rgb = torch.rand((N, S, C), generator=gen)
    if not training:
        rgb[3, 5, 0] = float("nan")
    he, hd, hr = g(e), g(dens), g(rgb)
    w_ref, _ = ops.weights_fwd(he, hd)
    c_ref, a_ref, m_ref, x_ref = ops.composite_fwd(hr, w_ref, he, training)
    for rep in range(2):  # the second call finds the scratch as the first one left it
        w, c, a, m, x = ops.render_fwd(he, hd, hr, training)
        for got, ref in ((w, w_ref), (c, c_ref), (a, a_ref), (m, m_ref), (x, x_ref)):
            assert torch.equal(got.nan_to_num(nan=-7.0), ref.nan_to_num(nan=-7.0))
    # the clip really is batch-global: expected depths lie inside [min midpoint, max midpoint] of the whole batch
    mid = (e[:, 1:] + e[:, :-1]) / 2
    assert float(x.min()) >= float(mid.min()) and float(x.max()) <= float(mid.max())
    w2, c2, a2, _, _ = ops.render_fwd(he, hd, hr, training, want_depth=False)
    assert torch.equal(w2, w_ref) and torch.equal(c2.nan_to_num(nan=-7.0), c_ref.nan_to_num(nan=-7.0)) and torch.equal(a2, a_ref)
    if not training:
        return
    gc = g(torch.rand((N, C), generator=gen) - 0.5)
    dw_loss = g(1e-3 * (torch.rand((N, S), generator=gen) - 0.5))
    dw = dw_loss.clone()
    drgb_ref = ops.composite_bwd(hr, w_ref, gc, dw)
    dd_ref = ops.weights_bwd(he, hd, w_ref, dw)
    keep = dw_loss.clone()
    drgb, dd = ops.render_bwd(he, hd, hr, w_ref, gc, dw_loss)
    assert torch.equal(drgb, drgb_ref) and torch.equal(dd, dd_ref)
    assert torch.equal(dw_loss, keep)  # read only


def test_train_losses_one_launch_equals_two():
    """tn_train_losses (+ tn_losses_finish) = tn_proposal_losses + tn_pixel_losses in one launch: same gradients bit for bit, same sums up to
    the order of the additions."""
    N = 512
    gen = torch.Generator().manual_seed(11)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    j0, j1, j2 = (torch.from_numpy(j) for j in synth.synth_jitters(N))
    sb, ws = [], []
    s_prev, w_prev = None, None
    for i, S in enumerate((256, 96, 48)):
        s = orc.spaced_bins(N, S, j0) if i == 0 else orc.pdf_resample(s_prev, w_prev, S, (j1, j2)[i - 1])
        e = orc.s_to_euclidean(s, nears, fars)
        smp = orc.Samples(s_bins=s, e_bins=e)
        w = orc.get_weights(smp.deltas, (5.0 * torch.rand((N, S, 1), generator=gen)) ** 3)
        sb.append(g(s)); ws.append(g(w[..., 0]))
        s_prev, w_prev = s, w
    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, N)
    img, is_th = synth.synth_gt(idx, cams)
    img, is_th = g(torch.from_numpy(img)), g(torch.from_numpy(is_th))
    pred = g(torch.rand((N, 4), generator=gen))

    def run(fused: bool):
        L = torch.zeros(16, device=DEV)
        Lp = torch.zeros((ops.LOSS_LINES, 16), device=DEV)
        d0, d1, d2, dp = torch.zeros_like(ws[0]), torch.zeros_like(ws[1]), torch.zeros_like(ws[2]), torch.zeros_like(pred)
        props = [(sb[0], ws[0], d0), (sb[1], ws[1], d1)]
        if fused:
            ops.train_losses(sb[2], ws[2], props, 0.002, 1.0, d2, Lp, pixel=(pred[:, :3], pred[:, 3:], img, is_th, 100.0, 1e-3, 1e-3, dp[:, :3], dp[:, 3:]))
            ops.losses_finish(Lp, L)
        else:
            ops.pixel_losses(pred[:, :3], pred[:, 3:], img, is_th, 100.0, 1e-3, 1e-3, L[0:8], dp[:, :3], dp[:, 3:])
            ops.proposal_losses(sb[2], ws[2], props, 0.002, 1.0, L[9:10], L[8:9], d2)
        return L, (d0, d1, d2, dp)

    La, ga = run(False)
    Lb, gb = run(True)
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
    assert float((La - Lb).abs().max()) <= 2e-6 * float(La.abs().max())
    assert float(Lb[4]) + float(Lb[5]) == N
    # the finisher also evaluates the camera regulariser (same arithmetic as tn_camera_reg) and ACCUMULATES into the loss vector
    pose = g(torch.from_numpy(synth.uniform("pose_f", (8, 6), -0.01, 0.01, SEED)))
    r1, g1 = torch.zeros(1, device=DEV), torch.zeros((8, 6), device=DEV)
    ops.camera_reg(pose, 1e-2, 1e-3, 1.0, r1, g1)
    L2, g2 = Lb.clone(), torch.zeros((8, 6), device=DEV)
    Lp = torch.zeros((ops.LOSS_LINES, 16), device=DEV)
    Lp[5, 9] = 0.25
    ops.losses_finish(Lp, L2, pose, 1e-2, 1e-3, 1.0, L2[11:12], g2)
    assert torch.equal(g1, g2) and float(L2[11]) == float(r1) and abs(float(L2[9]) - float(Lb[9]) - 0.25) <= 1e-7


@pytest.mark.parametrize("S,prop_grad", [(48, True), (48, False), (100, True), (200, True)])
def test_render_losses_bwd_is_the_three_launches(S, prop_grad):
    """tn_render_losses_bwd = tn_render_fwd(training) + tn_train_losses + tn_render_bwd in one launch: every rendered output and every gradient
    bit for bit (weights, composite, accumulation, both depths incl. the batch-wide clip, d composite, d weights of all three levels, d rgb,
    d density); the loss sums up to the order of their additions.  N = 2080 rays: more patches than blocks (the ray loop runs twice for some),
    accumulators that are NOT zero on entry, 48 samples (one per lane) and 100 / 200 (two / four per lane)."""
    N = 2080
    gen = torch.Generator().manual_seed(17)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    j0, j1, j2 = (torch.from_numpy(j) for j in synth.synth_jitters(N))
    sb, eb, ws = [], [], []
    s_prev, w_prev = None, None
    for i, Sl in enumerate((256, 96, S)):
        if i == 0:
            s_ = orc.spaced_bins(N, Sl, j0)
        else:
            s_ = orc.pdf_resample(s_prev, w_prev, Sl, (j1, j2)[i - 1])
        e = orc.s_to_euclidean(s_, nears, fars)
        w = orc.get_weights(orc.Samples(s_bins=s_, e_bins=e).deltas, (5.0 * torch.rand((N, Sl, 1), generator=gen)) ** 3)
        sb.append(g(s_)); eb.append(g(e)); ws.append(g(w[..., 0]))
        s_prev, w_prev = s_, w
    dens = g((5.0 * torch.rand((N, S), generator=gen)) ** 3 * 0.01)
    rgb = g(torch.rand((N, S, 4), generator=gen))
    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, N)
    img, is_th = synth.synth_gt(idx, cams)
    img, is_th = g(torch.from_numpy(img)), g(torch.from_numpy(is_th))
    acc0 = {k: g(torch.rand(shape, generator=gen) * 1e-3) for k, shape in (("d0", (N, 256)), ("d1", (N, 96)), ("d2", (N, S)), ("dc", (N, 4)))}

    def run(fused: bool):
        L = torch.zeros(16, device=DEV)
        Lp = torch.zeros((ops.LOSS_LINES, 16), device=DEV)
        d0, d1, d2, dc = (acc0[k].clone() for k in ("d0", "d1", "d2", "dc"))
        props = [(sb[0], ws[0], d0 if prop_grad else None), (sb[1], ws[1], d1 if prop_grad else None)]
        if fused:
            w, c, a, m, x, d_rgb, dd = ops.render_losses_bwd(eb[2], dens, rgb, sb[2], props, 0.002, 1.0, d2, img, is_th, 100.0, 1e-3, 1e-3, dc, Lp)
        else:
            w, c, a, m, x = ops.render_fwd(eb[2], dens, rgb, True)
            ops.train_losses(sb[2], w, props, 0.002, 1.0, d2, Lp, pixel=(c[:, :3], c[:, 3:], img, is_th, 100.0, 1e-3, 1e-3, dc[:, :3], dc[:, 3:]))
            d_rgb, dd = ops.render_bwd(eb[2], dens, rgb, w, dc, d2)
        ops.losses_finish(Lp, L)
        torch.cuda.synchronize()
        return L, dict(w=w, comp=c, acc=a, median=m, expected=x, d0=d0, d1=d1, d2=d2, d_comp=dc, d_rgb=d_rgb, d_density=dd)

    La, A = run(False)
    Lb, B = run(True)
    for k in A:
        assert torch.equal(A[k], B[k]), (k, float((A[k] - B[k]).abs().max()))
    assert float(A["d_density"].abs().max()) > 0 and float(A["d2"].sub(acc0["d2"]).abs().max()) > 0
    assert prop_grad == (float(A["d0"].sub(acc0["d0"]).abs().max()) > 0)
    assert float((La - Lb).abs().max()) <= 2e-6 * float(La.abs().max()), (La, Lb)
    assert float(Lb[4]) + float(Lb[5]) == N and all(float(Lb[k]) > 0 for k in (0, 1, 8, 9))
    with pytest.raises(RuntimeError):  # 2x2 patches: N must be a multiple of 4
        ops.render_losses_bwd(eb[2][:6], dens[:6], rgb[:6], sb[2][:6], [], 0.002, 1.0, acc0["d2"][:6].clone(), img[:6], is_th[:6], 100.0, 1e-3, 1e-3,
                              acc0["dc"][:6].clone(), torch.zeros((ops.LOSS_LINES, 16), device=DEV))


def test_interlevel_gradient_on_unsorted_bins_falls_back_to_the_full_walk():
    """The fast gradient path relies on sorted fine bins (lo_i, hi_i monotone); anything else takes the walk over every interval.  Checked
    against a direct evaluation of d wp_k = sum_i ([lo_i <= k <= hi_i] - [hi_i < k < lo_i]) g_i (csrc/tn_sampler.hip, interlevel_body)."""
    N, Sf, Sp = 6, 48, 96
    gen = torch.Generator().manual_seed(3)
    cp = torch.sort(torch.rand((N, Sp + 1), generator=gen), dim=1).values
    c = torch.sort(torch.rand((N, Sf + 1), generator=gen), dim=1).values
    c[1::2] = c[1::2][:, torch.randperm(Sf + 1, generator=gen)]  # every other ray: unsorted fine bins
    wf = torch.rand((N, Sf), generator=gen) * 0.05
    wp = torch.rand((N, Sp), generator=gen) * 0.01
    cy = torch.cat([torch.zeros(N, 1), torch.cumsum(wp.double(), 1).float()], 1).numpy()
    cpn, cn, wfn = cp.numpy(), c.numpy(), wf.numpy()
    ref = np.zeros((N, Sp), np.float64)
    total = 0.0
    scale = np.float32(1.0) / (np.float32(N) * np.float32(Sf))
    for r in range(N):
        for i in range(Sf):
            lo = int(np.clip(np.searchsorted(cpn[r, :-1], cn[r, i], side="right") - 1, 0, Sp - 1))
            hi = int(np.clip(np.searchsorted(cpn[r, 1:], cn[r, i + 1], side="right"), 0, Sp - 1))
            d = max(np.float32(wfn[r, i]) - (np.float32(cy[r, hi + 1]) - np.float32(cy[r, lo])), np.float32(0))
            total += float(d * d / (wfn[r, i] + np.float32(1e-7)))
            gi = float(np.float32(-2.0) * d / (wfn[r, i] + np.float32(1e-7)) * scale)
            if lo <= hi:
                ref[r, lo:hi + 1] += gi
            else:
                ref[r, hi + 1:lo] -= gi
    loss = torch.zeros(1, device=DEV)
    dw = torch.zeros((N, Sp), device=DEV)
    ops.interlevel_loss(g(c), g(wf), g(cp), g(wp), 1.0, loss, dw)
    assert abs(float(loss) - total * float(scale)) <= 1e-5 * total * float(scale)
    assert float((dw.cpu().double() - torch.from_numpy(ref)).abs().max()) <= 1e-5 * float(np.abs(ref).max())


def test_fused_first_and_last_launches_equal_the_separate_ones(golden_dir):
    """tn_sample_rays = tn_sample_pixels + tn_raygen, tn_pose_spaced_bins = tn_pose_apply_fwd + tn_spaced_bins (bit-identical);
    tn_pose_bwd_finish = tn_pose_apply_bwd + tn_camera_reg + tn_losses_finish (same sums up to the order of the atomic additions)."""
    gen = torch.Generator().manual_seed(21)
    cams = synth.synth_cameras()
    cam_t = {k: g(torch.from_numpy(np.ascontiguousarray(v))) for k, v in cams.items() if k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    Cn = cam_t["c2w"].shape[0]
    H, W = 20, 28
    images = [torch.rand((H + i, W + 2 * i, 3), generator=gen) for i in range(Cn)]
    cache = ops.ImageCache.build(images, torch.tensor([i % 2 for i in range(Cn)], dtype=torch.float32), torch.arange(Cn), DEV)
    N = 8 * 36
    u = g(torch.rand((N // 4, 3), generator=gen))
    idx, img, is_th, cam = ops.sample_pixels(cache, N, u, 2, want_camera_indices=True)
    o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t.get("distortion"))
    o2, d2, cam2, img2, is_th2, idx2 = ops.sample_rays(cache, N, u, cam_t, 2)
    for a, b in ((o, o2), (d, d2), (cam, cam2), (img, img2), (is_th, is_th2), (idx, idx2)):
        assert torch.equal(a, b)
    pose = g(torch.from_numpy(synth.uniform("pose_fl", (Cn, 6), -0.02, 0.02, SEED)))
    frozen = g(torch.tensor([0, 1] * (Cn // 2), dtype=torch.uint8))
    nears, fars, jit = g(torch.full((N,), 0.05)), g(torch.full((N,), 1000.0)), g(torch.rand(N, generator=gen))
    po, pd = ops.pose_apply_fwd(pose, frozen, cam, o, d)
    s, e = ops.spaced_bins(nears, fars, 256, jit)
    po2, pd2, s2, e2 = ops.pose_spaced_bins(pose, frozen, cam, o, d, nears, fars, 256, jit)
    for a, b in ((po, po2), (pd, pd2), (s, s2), (e, e2)):
        assert torch.equal(a, b)
    g_o, g_d = g(torch.rand((N, 3), generator=gen) - 0.5), g(torch.rand((N, 3), generator=gen) - 0.5)
    gp1, L1 = torch.zeros((Cn, 6), device=DEV), torch.zeros(16, device=DEV)
    Lp = g(torch.rand((ops.LOSS_LINES, 16), generator=gen))
    ops.pose_apply_bwd(pose, frozen, cam, d, g_o, g_d, gp1)
    ops.camera_reg(pose, 1e-2, 1e-3, 1.0, L1[11:12], gp1)
    ops.losses_finish(Lp, L1)
    gp2, L2 = torch.zeros((Cn, 6), device=DEV), torch.zeros(16, device=DEV)
    ops.pose_bwd_finish(pose, frozen, cam, d, g_o, g_d, gp2, 1e-2, 1e-3, 1.0, L2[11:12], Lp, L2)
    assert float((gp1 - gp2).abs().max()) <= 1e-6 * float(gp1.abs().max())
    assert float((L1 - L2).abs().max()) <= 1e-6 * float(L1.abs().max())
    assert float((L1[:11].cpu() - Lp.cpu().sum(0)[:11]).abs().max()) <= 1e-5
    gp3, L3 = torch.zeros((Cn, 6), device=DEV), torch.zeros(16, device=DEV)
    ops.pose_bwd_finish(pose, frozen, cam, d, g_o, g_d, gp3, 1e-2, 1e-3, 1.0, L3[12:13])  # second pose tensor of separate mode: no loss lines
    reg_only = torch.zeros(1, device=DEV)
    ops.camera_reg(pose, 1e-2, 1e-3, 1.0, reg_only, None)
    assert float((gp1 - gp3).abs().max()) <= 1e-6 * float(gp1.abs().max()) and float(L3[12]) == float(reg_only) and float(L3[:11].abs().max()) == 0.0


def test_pixel_losses_l1_camera_reg():
    cams = synth.synth_cameras()
    N = 512
    idx = synth.synth_ray_indices(cams, N)
    img, is_th = synth.synth_gt(idx, cams)
    img, is_th = torch.from_numpy(img), torch.from_numpy(is_th)
    pred = torch.from_numpy(synth.uniform("pred", (N, 4), 0.0, 1.0, SEED)).requires_grad_(True)
    ocfg = tiny_cfg("shared")
    pose = (torch.from_numpy(synth.uniform("pose_r", (8, 6), -0.01, 0.01, SEED))).requires_grad_(True)
    out = {"rgb": pred[:, :3], "rgb_thermal": pred[:, 3:]}
    ld = orc.loss_dict({"camera_optimizer.pose_adjustment": pose}, ocfg, out, img, is_th, training=False)
    reg = orc.camera_opt_regularizer(pose, ocfg, 1.0)
    (sum(ld.values()) + reg).backward()
    hp = g(pred.detach())
    dp = torch.zeros_like(hp)
    losses = torch.zeros(8, device=DEV)
    ops.pixel_losses(hp[:, :3], hp[:, 3:], g(img), g(is_th), 100.0, 1e-6, 1e-6, losses, dp[:, :3], dp[:, 3:])
    for i, k in enumerate(("rgb_loss", "thermal_loss", "tv_pixel_loss", "cross_channel_loss")):
        assert abs(float(losses[i]) - float(ld[k])) <= 2e-5 * abs(float(ld[k])) + 1e-12, k
    assert md(dp, pred.grad) <= 1e-5 * float(pred.grad.abs().max())
    lr = torch.zeros(1, device=DEV)
    gp = torch.zeros((8, 6), device=DEV)
    ops.camera_reg(g(pose.detach()), 1e-2, 1e-3, 1.0, lr, gp)
    assert abs(float(lr) - float(reg)) <= 1e-6 * float(reg)
    assert md(gp, pose.grad) <= 1e-6
    # l1 with the reference's detach asymmetry: a*|x.detach()-y| + b*|x-y.detach()|
    x = torch.from_numpy(synth.uniform("l1x", (1000,), 0.0, 3.0, SEED)).requires_grad_(True)
    y = torch.from_numpy(synth.uniform("l1y", (1000,), 0.0, 3.0, SEED)).requires_grad_(True)
    a, b = 5e-5, 5e-7
    l1 = torch.nn.functional.l1_loss
    ref = a * l1(x.detach(), y) + b * l1(x, y.detach())
    ref.backward()
    lo = torch.zeros(1, device=DEV)
    dx, dy = torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    ops.l1_loss(g(x.detach()), g(y.detach()), b, a, lo, dx, dy)
    assert abs(float(lo) - float(ref)) <= 1e-6 * float(ref)
    assert md(dx, x.grad) <= 1e-12 and md(dy, y.grad) <= 1e-12


def test_adam_matches_torch_optim():
    n = 100003
    p = torch.from_numpy(synth.uniform("ap", (n,), seed=SEED))
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-2, eps=1e-15)
    hp, m, v = g(p), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        grad = torch.from_numpy(synth.uniform(f"ag{step}", (n,), seed=SEED)) * 1e-3
        grad[::7] = 0.0
        ref_p.grad = grad.clone()
        opt.step()
        ops.adam_step(hp, g(grad), m, v, step, 1e-2)
        assert md(hp, ref_p) <= 2e-6, step
    # several optimiser groups (own step count / learning rate) of one arena in one launch == one launch per group, bit for bit
    bounds = [(0, 40000, 3, 1e-2), (40000, 40064, 1, 6e-4), (40064, 99996, 7, 2e-3), (99996, n, 2, 1e-3)]  # last range: 7 elements (tail only)
    grad = g(torch.from_numpy(synth.uniform("agr", (n,), seed=SEED)) * 1e-3)
    a = [t.clone() for t in (hp, m, v)]
    b = [t.clone() for t in (hp, m, v)]
    for lo, hi, st, lr in bounds:
        ops.adam_step(a[0][lo:hi], grad[lo:hi], a[1][lo:hi], a[2][lo:hi], st, lr)
    ops.adam_step_ranges(b[0], grad, b[1], b[2], bounds)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    with pytest.raises(RuntimeError):
        ops.adam_step_ranges(b[0], grad, b[1], b[2], [(2, 10, 1, 1e-3)])  # offsets must be multiples of 4


def test_adam_untouched_entries_are_skipped_exactly():
    """k_adam_ranges_amp leaves entries with g = m = v = +0 alone (no read of p, no write).  That must be indistinguishable from doing the
    arithmetic: bit for bit against k_adam_ranges (the same arithmetic without the shortcut) and against torch.optim.Adam to its usual 2e-6,
    over several steps with blocks of never-touched entries, entries whose gradient is zero in one step but whose moments are not (they must
    still move), a -0.0 gradient (not skipped: it is computed like any other), and zero_grads."""
    n = 4096 + 12
    p0 = torch.from_numpy(synth.uniform("ap0", (n,), seed=SEED))
    ref_p = p0.clone().to(DEV).requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-2, eps=1e-15)
    hp, m, v = g(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    fp, fm, fv = g(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)  # the full arithmetic on every entry
    dead = torch.zeros(n, dtype=torch.bool)
    dead[64:1024] = True  # whole float4 groups and whole cache lines never touched
    dead[2000:2003] = True  # part of a float4 group
    for step in range(1, 5):
        grad = torch.from_numpy(synth.uniform(f"agz{step}", (n,), seed=SEED)) * 1e-3
        grad[dead] = 0.0
        if step == 2:
            grad[1100:1300] = 0.0  # moments are non-zero here from step 1: these entries still move
            grad[1400] = -0.0
        ref_p.grad = g(grad).clone()
        opt.step()
        ops.adam_step_ranges(fp, g(grad), fm, fv, [(0, n, step, 1e-2)])
        hg = g(grad).clone()
        ops.adam_step_ranges_amp(hp, hg, m, v, [(0, n, step, 1e-2)], zero_grads=True)
        assert torch.equal(hp, fp) and torch.equal(m, fm) and torch.equal(v, fv), step
        assert md(hp, ref_p.detach()) <= 2e-6, step
        assert not hg.any(), "the launch consumes the gradients"
        assert torch.equal(hp.cpu()[dead], p0[dead])
        if step == 2:
            assert (hp.cpu()[1100:1300] != before[1100:1300]).all(), "zero gradient, non-zero moments: the entry still moves"
        before = hp.cpu().clone()


def test_empty_single_and_eval_chunk_sizes():
    """Edge sizes: N = 0 (valid, touches nothing), N = 1, and the reference's eval chunk of 32768 rays (8.4 M proposal points)."""
    from nerfstudio_thermal_amd.engine import RenderEngine

    ocfg, params, cfg, arena = setup_pair("shared")
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    z3 = torch.zeros((0, 3), device=DEV)
    s, e = ops.spaced_bins(torch.zeros(0, device=DEV), torch.zeros(0, device=DEV), 256)
    assert s.shape == (0, 257)
    d = ops.prop_density_fwd(eng.props[0], z3, z3, e)
    assert d.shape == (0, 256)
    w, _ = ops.weights_fwd(e, d)
    assert w.shape == (0, 256)
    # the fused entry points accept an empty batch too: whole-branch renders (inference and training), renderers, loss launch
    zc = torch.zeros(0, dtype=torch.int64, device=DEV)
    ze = torch.zeros(0, device=DEV)
    r0 = ops.render_rays_eval(eng.props, eng.field, z3, z3, zc, ze, ze, eng.counts, 1.0)
    assert r0["rgb"].shape == (0, 4) and r0["levels"][0]["weights"].shape == (0, 256)
    t0 = ops.render_rays_train(eng.props, eng.field, eng.pose, eng.frozen_rgb, z3, z3, zc, ze, ze, eng.counts, 1.0, None)
    assert t0["rgb"].shape == (0, 4) and t0["levels"][2]["e_bins"].shape == (0, 49)
    w0, c0, a0, m0, x0 = ops.render_fwd(torch.zeros((0, 49), device=DEV), torch.zeros((0, 48), device=DEV), torch.zeros((0, 48, 4), device=DEV), True)
    assert w0.shape == (0, 48) and c0.shape == (0, 4) and x0.shape == (0, 1)
    Lp = torch.zeros((ops.LOSS_LINES, 16), device=DEV)
    ops.train_losses(torch.zeros((0, 49), device=DEV), torch.zeros((0, 48), device=DEV), [(torch.zeros((0, 257), device=DEV), torch.zeros((0, 256), device=DEV), None)],
                     0.002, 1.0, None, Lp)
    torch.cuda.synchronize()
    assert float(Lp.abs().max()) == 0.0
    r1 = rays(1)
    out1, _ = eng.get_outputs(g(r1["origins"]), g(r1["directions"]), g(r1["camera_indices"]), training=False)
    with torch.no_grad():
        ref1 = orc.get_outputs(params, ocfg, r1["origins"], r1["directions"], r1["camera_indices"], training=False)
    assert md(out1["rgb"], ref1["rgb"]) <= 1e-3
    N = 32768
    r = rays(N)
    out, br = eng.get_outputs(g(r["origins"]), g(r["directions"]), g(r["camera_indices"]), training=False)
    assert out["rgbt"].shape == (N, 4) and bool(torch.isfinite(out["rgbt"]).all())
    sub = slice(1000, 1064)
    with torch.no_grad():
        ref = orc.get_outputs(params, ocfg, r["origins"][sub], r["directions"][sub], r["camera_indices"][sub], training=False)
    assert md(out["rgbt"][sub], torch.cat([ref["rgb"], ref["rgb_thermal"]], -1)) <= 1e-3


def test_sample_pixels_matches_reference_golden(golden_dir):
    """N2: tn_sample_pixels == the reference's PatchPixelSampler on the jagged RGB + thermal batch (bit-exact: indices and gathered pixels),
    then the full data path sample -> raygen against the oracle, and the edge cases of the split over images."""
    from helpers import pixel_batch

    b = pixel_batch(golden_dir)
    cache = ops.ImageCache.build(b["images"], b["is_thermal"], b["image_idx"], DEV)
    idx, img, is_th, cam_idx = ops.sample_pixels(cache, b["num_rays"], g(b["u"]), want_camera_indices=True)
    assert torch.equal(cam_idx, idx[:, 0])
    assert torch.equal(idx.cpu(), b["ref"]["indices"])
    assert torch.equal(img.cpu(), b["ref"]["image"])
    assert torch.equal(is_th.cpu(), b["ref"]["is_thermal"])
    # sampled indices feed the ray generator exactly like the host-side batch does
    cams = synth.synth_cameras()
    t = lambda k: torch.from_numpy(cams[k])  # noqa: E731
    o, d, _, _ = ops.raygen(idx, *(g(t(k)) for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")))
    ro, rd, _, _ = orc.generate_rays(b["ref"]["indices"], t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
    assert md(o, ro) <= 1e-6 and md(d, rd) <= 2e-6
    # larger / ragged splits against the oracle: the last image takes the remainder; patch sizes 1, 2, 4
    for n, ps in ((4096, 2), (8 * 36 + 12, 2), (8 * 7 + 3, 1), (8 * 64, 4)):
        u = torch.from_numpy(synth.synth_patch_uniforms(n // (ps * ps), seed=n))
        want = orc.sample_pixels(b["images"], b["is_thermal"], b["image_idx"], n, u, ps)
        got = ops.sample_pixels(cache, n, g(u), ps)
        for a, w in zip(got, want):
            assert torch.equal(a.cpu(), w), (n, ps)
    with pytest.raises(RuntimeError):  # 8*4+2 rays do not split into whole 2x2 patches
        ops.sample_pixels(cache, 34, g(torch.zeros((8, 3))), 2)


def test_renderer_properties_of_the_reference_suite():
    """The property checks of the reference's tests/model_components/test_renderers.py:11-82 (uniform weights over 10 samples: max(rgb) > 0.9,
    zero colours -> 0, accumulation > 0.9, median / expected depth > 0), run through tn_composite_fwd / tn_weights_fwd for RGB, RGBT and
    1-channel renders."""
    S = 10
    w = torch.full((3, S), 1.0 / S, device=DEV)
    e = torch.linspace(0.0, 101.0, S + 1, device=DEV).repeat(3, 1).contiguous()
    for Cn in (3, 4, 1):
        ones = torch.ones((3, S, Cn), device=DEV)
        for training in (True, False):
            comp, acc, med, exp = ops.composite_fwd(ones, w, e, training)
            assert float(comp.max()) > 0.9 and float(acc.max()) > 0.9
            assert float(med.min()) > 0 and float(exp.min()) > 0
            comp0, _, _, _ = ops.composite_fwd(ones * 0, w, e, training)
            assert float(comp0.abs().max()) <= 1e-6
    # weights from densities: a dense slab gives (almost) all weight to the first samples and the weights sum to < = 1
    dens = torch.full((3, S), 5.0, device=DEV)
    hw, med = ops.weights_fwd(e, dens, want_median=True)
    assert float(hw.sum(1).max()) <= 1.0 + 1e-6 and float(hw[:, 0].min()) > 0.99 and float(med.min()) > 0


def test_out_of_box_and_degenerate_rays_match_oracle():
    """Selector / contraction edge cases: origins far outside the scene box, positions exactly on the contraction boundary (|x|_inf = 1),
    a zero direction (every sample at the origin of the ray), a ray parked at 1e9 (contracted radius rounds to 2 -> unit-cube face -> selector off)."""
    ocfg, params, cfg, arena = setup_pair()
    N, S = 8, 96
    o = torch.tensor([[0.0, 0.0, 0.0], [5.0, -7.0, 3.0], [1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [100.0, 100.0, 100.0], [0.3, 0.2, -0.9], [0.0, 0.0, 1e-30], [1e9, -1e9, 3e8]])
    d = torch.tensor([[0.0, 0.0, 1.0], [-0.5, 0.7, -0.3], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], [-1.0, -1.0, -1.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    d = torch.where(d.norm(dim=1, keepdim=True) > 0, d / d.norm(dim=1, keepdim=True).clamp_min(1e-30), d)
    nears, fars = torch.zeros(N, 1), torch.full((N, 1), 1000.0)
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    pos = smp.positions(o, d)
    with torch.no_grad():
        ref_p = orc.prop_density(params, "proposal_networks", 1, ocfg, pos)
        ref_f, _, _, _ = orc.field_density(params, "field", ocfg, pos)
    hp = ops.prop_density_fwd(prop_params(arena, "proposal_networks", 1, cfg), g(o), g(d), g(e))
    hf = ops.field_density_fwd(field_params(arena, "field", cfg), g(o), g(d), g(e))
    for got, ref, name in ((hp, ref_p[..., 0], "proposal"), (hf, ref_f[..., 0], "field")):
        ref = ref.to(DEV)
        same_zero = (got == 0) == (ref == 0)
        assert bool(same_zero.all()), (name, int((~same_zero).sum()))  # the selector switches off exactly the same samples
        rel = ((got - ref).abs() / ref.abs().clamp_min(1e-6)).max()
        assert float(rel) <= 1e-4, (name, float(rel))
    assert float((hp[7] == 0).float().mean()) == 1.0 and float((hp[:7] == 0).float().mean()) == 0.0  # only the ray at 1e9 is switched off


def test_non_finite_inputs_follow_torch_nan_to_num():
    """RaySamples.get_weights ends in torch.nan_to_num (cameras/rays.py:148) and the eval renderers sanitise the colours
    (model_components/renderers.py:118-120): infinite / huge densities and NaN colours must come out as torch produces them."""
    N, S = 6, 48
    nears, fars = torch.full((N, 1), 0.05), torch.full((N, 1), 1000.0)
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    dens = torch.from_numpy(synth.uniform("nf_dens", (N, S, 1), 0.0, 2.0, SEED))
    dens[0, 5] = float("inf")
    dens[1, :] = 3.0e38
    dens[2, 10] = float("inf"); dens[2, 11] = float("inf")
    dens[3, 0] = float("inf")
    dens[4, :] = 0.0
    w_ref = orc.get_weights(smp.deltas, dens)
    assert bool(torch.isfinite(w_ref).all())
    hw, _ = ops.weights_fwd(g(e), g(dens[..., 0]))
    assert bool(torch.isfinite(hw).all()) and md(hw, w_ref[..., 0]) <= 2e-6
    rgb = torch.from_numpy(synth.uniform("nf_rgb", (N, S, 4), 0.0, 1.0, SEED))
    rgb[0, 3, 1] = float("nan"); rgb[2, 7, 0] = float("inf"); rgb[5, :, 3] = float("-inf")
    ref = orc.composite_rgb(rgb, w_ref, training=False)  # eval path: nan_to_num before, clamp after
    comp, acc, _, _ = ops.composite_fwd(g(rgb), hw, g(e), training=False)
    assert bool(torch.isfinite(comp).all()) and md(comp, ref) <= 1e-5


def test_device_data_manager_batches(golden_dir):
    """DeviceDataManager.next_train == sample_pixels + raygen on the uniforms it draws (with and without the one-step-ahead prefetch), fresh
    pixels every call, ground truth gathered from the cached images."""
    from helpers import pixel_batch
    from nerfstudio_thermal_amd.data import DeviceDataManager

    b = pixel_batch(golden_dir)
    cache = ops.ImageCache.build(b["images"], b["is_thermal"], b["image_idx"], DEV)
    cams = synth.synth_cameras()
    cam_t = {k: g(torch.from_numpy(cams[k])) for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    for prefetch in (False, True, "cowork"):  # ("cowork" without a training step in between: every batch is launched just before it is handed out)
        torch.manual_seed(11)
        dm = DeviceDataManager(cache, cam_t, 256, 2, prefetch=prefetch)
        got = [dm.next_train(i) for i in range(3)]
        torch.cuda.synchronize()
        torch.manual_seed(11)
        pool = ops.UniformPool(DEV)  # the manager draws its uniforms through one of these (one torch.rand per 32 batches)
        for o, d, cam, img, is_th in got:
            u = pool.take((64, 3))
            idx, img_r, th_r, cam_r = ops.sample_pixels(cache, 256, u, 2, want_camera_indices=True)
            o_r, d_r, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
            for a, r in ((o, o_r), (d, d_r), (cam, cam_r), (img, img_r), (is_th, th_r)):
                assert torch.equal(a, r), prefetch
        ops.flush_pending_sample()  # (the batch the "cowork" manager prepared last: nobody will take it)
        assert not torch.equal(got[0][2], got[1][2]) or not torch.equal(got[0][3], got[1][3])  # a new batch every call


def test_field_fwd_bwd_many_cameras():
    """The per-camera tables of the training path (camhead in the forward, the per-camera sums and their finishing jobs in the backward) with a
    few hundred cameras in random order over the rays: every tile of 32 samples may hold a camera boundary, cameras without a ray in the batch
    exist, the finishing jobs outnumber the eight embedding-column jobs, and a second backward on the same forward adds the same gradients again
    (the sums are cleared by the finishing launch, not only by the next forward)."""
    I = 300
    ocfg, params, cfg, arena = setup_pair("shared", num_images=I, is_thermal_cam=tuple([0, 1] * (I // 2)))
    N, S = 257, 48
    r = rays(N)
    gen = torch.Generator().manual_seed(5)
    cam = torch.randint(0, I - 40, (N,), generator=gen)  # (the last 40 cameras see no ray)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    smp = orc.Samples(s_bins=s, e_bins=e)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    dens, geo, pre, enc = orc.field_density(p, "field", ocfg, smp.positions(r["origins"], r["directions"]))
    rgb = orc.field_color(p, "field", ocfg, r["directions"], geo, cam, True)
    fld = field_params(arena, "field", cfg, with_grads=True)
    hd, hrgb, _ = ops.field_fwd(fld, g(r["origins"]), g(r["directions"]), g(cam), g(e), True, want_pre=True)
    assert md(hd, dens[..., 0]) <= 1e-4 and md(hrgb, rgb) <= 1e-4
    C = fld.num_channels
    gd = torch.from_numpy(synth.uniform("gfd", (N, S, 1), seed=SEED))
    gc = torch.from_numpy(synth.uniform("gfc", (N, S, C), seed=SEED))
    ((dens * gd).sum() + (rgb * gc).sum()).backward()
    arena.zero_grad()
    k = orc.field_keys("field")
    for rep in (1, 2):
        ops.field_bwd(fld, g(r["origins"]), g(r["directions"]), g(cam), g(e), g(gd[..., 0]), g(gc), None, None)
        for short in ("table", "w0", "b0", "w1", "b1", "hw0", "hb0", "hw1", "hb1", "hw2", "hb2", "emb"):
            ref = p[k[short]].grad * rep
            got = arena.grad_view(k[short])
            scale = float(ref.abs().max())
            assert md(got, ref) <= 3e-4 * scale, (rep, short, md(got, ref), scale)
    gemb = arena.grad_view(k["emb"])
    assert float(gemb[I - 40:].abs().max()) == 0.0  # cameras without a ray keep an exactly-zero embedding gradient


@pytest.mark.parametrize("shape", ["main", "proposal"])
def test_hash_scatter_paths_agree(monkeypatch, shape):
    """TN_SCATTER_MODE=2 (segmented: block-private record regions, the default), 1 (binned: per-bucket arrays with reservations) and 0 (atomics +
    dense replicas) are three ways to the same sums: table gradient and d origins / d directions at the production shapes of the main grid
    (one fold block per bucket) and of the first proposal grid (buckets cut into chunks that flush with float atomics)."""
    L, log2T, N, S, max_res = (16, 19, 4096, 48, 2048) if shape == "main" else (5, 17, 4096, 256, 128)
    res = ops.level_resolutions(L, 16, max_res)
    r = rays(N)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    table = torch.from_numpy(synth.uniform("hst", (L * 2**log2T, 2), seed=SEED)).to(DEV)
    g_enc = g(torch.from_numpy(synth.uniform("hsg", (L, N * S, 2), seed=SEED)) * 1e-3)
    out = {}
    for mode in ("1", "2", "0"):
        monkeypatch.setenv("TN_SCATTER_MODE", mode)
        for zero in (False, True):
            tg = torch.zeros((L * 2**log2T, 2), device=DEV)
            d_o, d_d = torch.zeros((N, 3), device=DEV), torch.zeros((N, 3), device=DEV)
            ops.hash_scatter(table, tg, L, log2T, res, g(r["origins"]), g(r["directions"]), g(e), g_enc, d_o, d_d, grad_is_zero=zero)
            torch.cuda.synchronize()
            out[mode, zero] = (tg, d_o, d_d)
    ref = out["1", False]
    scale = [float(t.abs().max()) for t in ref]
    assert min(scale) > 0
    for key, got in out.items():
        assert torch.equal(got[0] == 0, ref[0] == 0), key
        for t, rt, sc, name in zip(got, ref, scale, ("table_grad", "d_origins", "d_directions")):
            # double-precision sums per bucket rounded once (modes 1, 2) against float atomics (mode 0, chunk flushes, the d position sums)
            assert float((t - rt).abs().max()) <= (1e-5 if name == "table_grad" else 1e-4) * sc, (key, name, float((t - rt).abs().max()), sc)


def test_hash_scatter_store_variant_equals_the_adding_one():
    """TnGrid.table_grad_is_zero: with the promise kept (zeros in table_grad) the storing fold gives what the adding one gives -- same non-zero
    pattern, same values up to the order of the fold's double-precision atomics -- at the production shape of the main grid."""
    L, log2T = 16, 19
    N, S = 4096, 48
    res = ops.level_resolutions(L, 16, 2048)
    r = rays(N)
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    s, e = sample_level(N, S, nears, fars)
    table = torch.from_numpy(synth.uniform("hst", (L * 2**log2T, 2), seed=SEED)).to(DEV)
    g_enc = g(torch.from_numpy(synth.uniform("hsg", (L, N * S, 2), seed=SEED)) * 1e-3)
    out = []
    for zero in (False, True):
        tg = torch.zeros((L * 2**log2T, 2), device=DEV)
        ops.hash_scatter(table, tg, L, log2T, res, g(r["origins"]), g(r["directions"]), g(e), g_enc, None, None, grad_is_zero=zero)
        out.append(tg)
    a, b = out
    assert torch.equal(a == 0, b == 0)
    scale = float(a.abs().max())
    assert float((a - b).abs().max()) <= 1e-5 * scale  # (float atomics of the overflow / replica paths, order of the fold's double atomics)


@pytest.mark.parametrize("scatter_mode", ["2", "1"])
def test_field_bwd_d_position_as_co_work_of_the_bin_launch(monkeypatch, scatter_mode):
    """The main field's d position pass runs in extra blocks of the table scatter's bin launch (tn_field_dpos.h; segmented path) or as a launch of
    its own (TN_DPOS_COWORK=0, and always on the binned path): the same d origins / d directions up to the order of their float atomics, and
    every other gradient of the field bit for bit (the embedding's finisher rides at the head of the pass either way)."""
    ocfg, params, cfg, arena = setup_pair("shared")
    N, S = 515, 48
    r = rays(N)
    cam = torch.arange(N) % ocfg.num_images
    nears, fars = torch.ones(N, 1) * 0.05, torch.ones(N, 1) * 1000.0
    _, e = sample_level(N, S, nears, fars)
    fld = field_params(arena, "field", cfg, with_grads=True)
    gd = torch.from_numpy(synth.uniform("gpd", (N, S), seed=SEED))
    gc = torch.from_numpy(synth.uniform("gpc", (N, S, fld.num_channels), seed=SEED))
    k = orc.field_keys("field")
    monkeypatch.setenv("TN_SCATTER_MODE", scatter_mode)
    res = {}
    for cw in ("1", "0"):
        monkeypatch.setenv("TN_DPOS_COWORK", cw)
        arena.zero_grad()
        d_o, d_d = torch.zeros((N, 3), device=DEV), torch.zeros((N, 3), device=DEV)
        ops.field_fwd(fld, g(r["origins"]), g(r["directions"]), g(cam), g(e), True)
        ops.field_bwd(fld, g(r["origins"]), g(r["directions"]), g(cam), g(e), g(gd), g(gc), d_o, d_d)
        torch.cuda.synchronize()
        res[cw] = {s_: arena.grad_view(k[s_]).detach().clone() for s_ in ("table", "w0", "b0", "w1", "b1", "hw0", "hb0", "hw1", "hb1", "hw2", "hb2", "emb")}
        res[cw]["d_o"], res[cw]["d_d"] = d_o, d_d
    for name, ref in res["0"].items():
        scale = float(ref.abs().max())
        assert scale > 0, name
        if name in ("d_o", "d_d"):
            assert md(res["1"][name], ref) <= 1e-5 * scale, (name, md(res["1"][name], ref), scale)
        elif name == "table":
            assert torch.equal(res["1"][name] == 0, ref == 0) and md(res["1"][name], ref) <= 1e-6 * scale, name  # (order of the fold's double atomics)
        else:
            assert md(res["1"][name], ref) <= 2e-6 * scale, (name, md(res["1"][name], ref), scale)

