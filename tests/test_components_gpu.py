"""The named API of BASELINE's north_star, each class driven STANDALONE against the oracle: ProposalNetworkSampler(ray_bundle, density_fns),
UniformLinDispPiecewiseSampler, PDFSampler, RGBRenderer / RGBTRenderer / DepthRenderer / AccumulationRenderer, HashMLPDensityField.density_fn
and ThermalNerfactoField.forward (model_components/ray_samplers.py:523-618, renderers.py:214,394,486,528, fields/density_fields.py:95-118,
fields/thermal_nerfacto_field.py:91-99)."""
import numpy as np
import pytest
import torch

import thermal_nerfacto_oracle as orc
from helpers import make_params, tiny_cfg, SEED
from nerfstudio_thermal_amd import synth
from nerfstudio_thermal_amd.model_components import (AccumulationRenderer, DepthRenderer, FieldHeadNames, NearFarCollider, PDFSampler,
                                                     ProposalNetworkSampler, RGBRenderer, RGBTRenderer, UniformLinDispPiecewiseSampler)
from nerfstudio_thermal_amd.rays import RayBundle
from test_hip_ops_gpu import md, outlier_fraction, pkg_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"
N = 96


def build_model(mode="shared"):
    from nerfstudio_thermal_amd.model import SceneBox

    ocfg = tiny_cfg(mode)
    cfg = pkg_cfg(ocfg)
    model = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=ocfg.num_images,
                      metadata={"is_thermal": list(ocfg.is_thermal_cam)}, device=DEV)
    params = make_params(ocfg)
    model.arena.load(params)
    return ocfg, params, model


def ray_bundle(training):
    r = {k: torch.from_numpy(v) for k, v in synth.synth_rays_simple(N, 11).items()}
    rb = RayBundle(origins=r["origins"].to(DEV), directions=r["directions"].to(DEV), pixel_area=torch.ones(N, 1, device=DEV),
                   camera_indices=r["camera_indices"][:, None].to(DEV))
    col = NearFarCollider(near_plane=0.05, far_plane=1000.0)
    col.train(training)
    return r, col(rb)


def bins(rs):
    s = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], -1)
    e = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[..., -1:, 0]], -1)
    return s, e


@pytest.mark.parametrize("training", [False, True])
def test_spaced_and_pdf_samplers_standalone(training):
    r, rb = ray_bundle(training)
    j0, j1, _ = (torch.from_numpy(j) for j in synth.synth_jitters(N))
    nears = torch.full((N, 1), 0.05 if training else 0.0)
    fars = torch.full((N, 1), 1000.0)
    s0 = UniformLinDispPiecewiseSampler(single_jitter=True)
    s0.train(training)
    rs0 = s0(rb, num_samples=256, jitter=j0.to(DEV).reshape(-1) if training else None)
    ref_s0 = orc.spaced_bins(N, 256, j0 if training else None)
    hs, he = bins(rs0)
    assert md(hs, ref_s0) == 0.0
    assert md(he, orc.s_to_euclidean(ref_s0, nears, fars)) <= 1e-6 * 1000.0
    assert rs0.shape == (N, 256) and rs0.frustums.origins.shape == (N, 256, 3) and rs0.frustums.origins.stride(1) == 0  # broadcast view along the samples
    assert rs0.frustums.starts.shape == (N, 256, 1) and rs0.camera_indices.shape == (N, 256, 1)
    # RaySamples.get_weights + PDFSampler on synthetic densities
    dens = torch.from_numpy(synth.uniform("cmp_dens", (N, 256, 1), 0.0, 40.0, SEED))
    w = rs0.get_weights(dens.to(DEV))
    smp = orc.Samples(s_bins=ref_s0, e_bins=orc.s_to_euclidean(ref_s0, nears, fars))
    ref_w = orc.get_weights(smp.deltas, dens)
    assert md(w, ref_w) <= 2e-6
    pdf = PDFSampler(include_original=False, single_jitter=True)
    pdf.train(training)
    rs1 = pdf(rb, rs0, w, num_samples=96, jitter=j1.to(DEV).reshape(-1) if training else None)
    ref_s1 = orc.pdf_resample(ref_s0, ref_w, 96, j1 if training else None)
    assert outlier_fraction(bins(rs1)[0], ref_s1, 2e-6) <= 0.005
    # spacing_to_euclidean_fn closure of the returned samples
    assert md(rs1.spacing_to_euclidean_fn(bins(rs1)[0]), bins(rs1)[1]) <= 1e-3


def test_proposal_network_sampler_standalone_eval():
    """ProposalNetworkSampler(ray_bundle, density_fns) -> (RaySamples, weights_list, ray_samples_list) against oracle.proposal_sample."""
    ocfg, params, model = build_model()
    model.eval()
    r, rb = ray_bundle(False)
    sampler = ProposalNetworkSampler(num_nerf_samples_per_ray=48, num_proposal_samples_per_ray=(256, 96), num_proposal_network_iterations=2,
                                     single_jitter=True)
    sampler.eval()
    rs, weights_list, rs_list = sampler(rb, model.density_fns)
    nears, fars = torch.zeros(N, 1), torch.full((N, 1), 1000.0)
    with torch.no_grad():
        ref_s, ref_w, ref_list = orc.proposal_sample(params, ocfg, "proposal_networks", r["origins"], r["directions"], nears, fars, 1.0, None)
    assert len(weights_list) == 2 and len(rs_list) == 2 and weights_list[0].shape == (N, 256, 1) and weights_list[1].shape == (N, 96, 1)
    for i in range(2):
        assert outlier_fraction(bins(rs_list[i])[0], ref_list[i].s_bins, 2e-6) <= 0.01, i
        assert outlier_fraction(weights_list[i], ref_w[i], 1e-5) <= 0.01, i
    assert outlier_fraction(bins(rs)[0], ref_s.s_bins, 2e-6) <= 0.01
    # set_anneal / step_cb bookkeeping (ray_samplers.py:568-575)
    sampler.set_anneal(0.5); sampler.step_cb(7)
    assert sampler._anneal == 0.5 and sampler._step == 7 and sampler._steps_since_update == 1


def test_density_fn_and_field_forward_standalone():
    ocfg, params, model = build_model()
    model.eval()
    # HashMLPDensityField.density_fn at explicit positions (Field.density_fn, fields/base_field.py:48-68)
    pos = torch.from_numpy(synth.uniform("cmp_pos", (50, 7, 3), -2.5, 2.5, SEED))
    for i in range(2):
        got = model.proposal_networks[i].density_fn(pos.to(DEV))
        with torch.no_grad():
            ref = orc.prop_density(params, "proposal_networks", i, ocfg, pos)
        assert got.shape == (50, 7, 1)
        assert md(got / ref.clamp_min(1e-6).to(DEV), ref / ref.clamp_min(1e-6)) <= 5e-5, i
    # ThermalNerfactoField.forward on RaySamples built by the bundle
    r, rb = ray_bundle(False)
    s = orc.spaced_bins(N, 48, None)
    e = orc.s_to_euclidean(s, torch.zeros(N, 1), torch.full((N, 1), 1000.0))
    rs = rb.get_ray_samples(bin_starts=e[:, :-1, None].to(DEV), bin_ends=e[:, 1:, None].to(DEV), spacing_starts=s[:, :-1, None].to(DEV),
                            spacing_ends=s[:, 1:, None].to(DEV))
    rs.e_bins, rs.s_bins = e.to(DEV).contiguous(), s.to(DEV).contiguous()
    out = model.field(rs)
    smp = orc.Samples(s_bins=s, e_bins=e)
    with torch.no_grad():
        dens, geo, _, _ = orc.field_density(params, "field", ocfg, smp.positions(r["origins"], r["directions"]))
        rgb = orc.field_color(params, "field", ocfg, r["directions"], geo, r["camera_indices"], False)
    assert set(out) == {FieldHeadNames.RGB, FieldHeadNames.DENSITY}
    assert md(out[FieldHeadNames.DENSITY], dens) <= 1e-4 and md(out[FieldHeadNames.RGB], rgb) <= 1e-4
    d_only, _ = model.field.get_density(rs)
    assert md(d_only, dens) <= 1e-4


@pytest.mark.parametrize("training", [False, True])
def test_renderers_standalone(training):
    S = 48
    s = orc.spaced_bins(N, S, None)
    e = orc.s_to_euclidean(s, torch.full((N, 1), 0.05), torch.full((N, 1), 1000.0))
    smp = orc.Samples(s_bins=s, e_bins=e)
    dens = torch.from_numpy(synth.uniform("cmp_rd", (N, S, 1), 0.0, 3.0, SEED)) ** 3
    w = orc.get_weights(smp.deltas, dens)
    r, rb = ray_bundle(training)
    rs = rb.get_ray_samples(bin_starts=e[:, :-1, None].to(DEV), bin_ends=e[:, 1:, None].to(DEV))
    rs.e_bins = e.to(DEV).contiguous()
    for cls, C in ((RGBRenderer, 3), (RGBTRenderer, 4)):
        rgb = torch.from_numpy(synth.uniform(f"cmp_rgb{C}", (N, S, C), -0.2, 1.2, SEED))
        rgb[3, 5, 0] = float("nan")  # eval-mode nan_to_num (renderers.py:118-133)
        ren = cls()
        ren.train(training)
        got = ren(rgb.to(DEV), w.to(DEV))
        ref = orc.composite_rgb(rgb, w, training)
        ok = ~torch.isnan(ref).any(dim=-1)
        assert got.shape == (N, C) and md(got[ok.to(DEV)], ref[ok]) <= 2e-6
    acc = AccumulationRenderer()(w.to(DEV))
    assert md(acc, orc.accumulation(w)) <= 2e-6
    med = DepthRenderer(method="median")(w.to(DEV), rs)
    assert outlier_fraction(med, orc.depth_median(w, smp), 1e-5) <= 0.02
    exp = DepthRenderer(method="expected")(w.to(DEV), rs)
    assert md(exp, orc.depth_expected(w, smp)) <= 1e-4
