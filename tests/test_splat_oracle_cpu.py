"""N4 oracle (oracle/splat_oracle.py, parity UNPINNED: gsplat is outside the reference tree) -- known-answer checks of the restated
algorithm itself: closed-form cases of EWA projection, SH colour and front-to-back blending."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import splat_oracle as so  # noqa: E402


def one_gaussian(mean, log_scale=-2.0, opacity_logit=10.0, dc=(1.0, 0.5, 0.0), dc_t=0.7):
    z = lambda *s: torch.zeros(*s)  # noqa: E731
    return {"means": torch.tensor([mean], dtype=torch.float32), "scales": torch.full((1, 3), log_scale), "quats": torch.tensor([[1.0, 0, 0, 0]]),
            "opacities": torch.tensor([[opacity_logit]]), "features_dc": torch.tensor([dc]), "features_rest": z(1, 15, 3),
            "features_dc_thermal": torch.tensor([[dc_t]]), "features_rest_thermal": z(1, 15, 1)}


def cam():
    return so.look_at_camera((3.0, 0.0, 0.0)), 100.0, 100.0, 32.0, 24.0, 64, 48


def test_projection_of_a_centred_isotropic_gaussian():
    c2w, fx, fy, cx, cy, W, H = cam()
    viewmat, proj = so.camera_matrices(c2w, fx, fy, W, H)
    s = math.exp(-2.0)
    pj = so.project_gaussians(torch.zeros(1, 3), torch.full((1, 3), s), 1.0, torch.tensor([[1.0, 0, 0, 0]]), viewmat, proj, fx, fy, cx, cy, H, W)
    # on the optical axis at depth 3: centre = principal point - 0.5, cov2d = (f s / z)^2 I + 0.3 I
    assert torch.allclose(pj["xys"], torch.tensor([[cx - 0.5, cy - 0.5]]), atol=1e-4)
    assert abs(float(pj["depths"]) - 3.0) < 1e-5
    var = (fx * s / 3.0) ** 2 + 0.3
    assert torch.allclose(pj["conics"], torch.tensor([[1 / var, 0.0, 1 / var]]), atol=1e-4)
    assert int(pj["radii"]) == math.ceil(3.0 * math.sqrt(var))
    assert abs(float(pj["compensation"]) - ((var - 0.3) / var)) < 1e-4  # sqrt(det_orig / det_blur) for an isotropic footprint
    # behind the camera / inside the near clip: culled
    pj2 = so.project_gaussians(torch.tensor([[5.0, 0, 0]]), torch.full((1, 3), s), 1.0, torch.tensor([[1.0, 0, 0, 0]]), viewmat, proj, fx, fy, cx, cy, H, W)
    assert int(pj2["radii"]) == 0 and int(pj2["num_tiles_hit"]) == 0


def test_sh_degree_zero_is_the_dc_colour_and_degree_one_follows_the_view_direction():
    d = torch.tensor([[0.0, 0.0, 1.0]])
    co = torch.zeros(1, 16, 3)
    co[0, 0] = torch.tensor([1.0, 2.0, 3.0])
    assert torch.allclose(so.spherical_harmonics(0, d, co), so.SH_C0 * co[:, 0])
    co[0, 2] = torch.tensor([1.0, 0.0, 0.0])  # the z-linear band
    out = so.spherical_harmonics(1, d * 5.0, co)  # directions are normalised inside
    assert torch.allclose(out, so.SH_C0 * co[:, 0] + torch.tensor([[so.SH_C1, 0.0, 0.0]]), atol=1e-6)


def test_single_opaque_gaussian_centre_pixel_and_background():
    c2w, fx, fy, cx, cy, W, H = cam()
    p = one_gaussian((0.0, 0.0, 0.0))
    out = so.render(p, c2w, fx, fy, cx, cy, W, H, sh_degree_to_use=0, background=torch.tensor([0.1, 0.2, 0.3]), background_thermal=0.05)
    col = torch.clamp(so.SH_C0 * torch.tensor([1.0, 0.5, 0.0]) + 0.5, min=0.0)
    var = (fx * math.exp(-2.0) / 3.0) ** 2 + 0.3
    # pixel (iy=23, ix=31) has its centre at (31.5, 23.5) = the Gaussian's centre: sigma = 0, alpha = min(0.999, sigmoid(10))
    a = min(0.999, 1 / (1 + math.exp(-10.0)))
    want = a * col + (1 - a) * torch.tensor([0.1, 0.2, 0.3])
    assert torch.allclose(out["rgb"][23, 31], want, atol=1e-5)
    assert abs(float(out["thermal"][23, 31]) - (a * (so.SH_C0 * 0.7 + 0.5) + (1 - a) * 0.05)) < 1e-5
    assert abs(float(out["accumulation"][23, 31]) - a) < 1e-6
    assert abs(float(out["depth"][23, 31]) - 3.0) < 1e-4
    # one pixel to the right: sigma = 0.5 / var
    a1 = min(0.999, (1 / (1 + math.exp(-10.0))) * math.exp(-0.5 / var))
    assert abs(float(out["accumulation"][23, 32]) - a1) < 1e-5
    # far corner: untouched -> background, accumulation 0, depth = max of the un-normalised depth image
    assert torch.allclose(out["rgb"][0, 0], torch.tensor([0.1, 0.2, 0.3])) and float(out["accumulation"][0, 0]) == 0.0
    assert float(out["depth"][0, 0]) == pytest.approx(float((out["depth"] * out["accumulation"]).max()), rel=1e-5)


def test_two_gaussians_blend_front_to_back_and_order_is_by_depth_not_by_index():
    c2w, fx, fy, cx, cy, W, H = cam()
    near, far = one_gaussian((1.0, 0.0, 0.0), opacity_logit=0.0, dc=(1.0, 1.0, 1.0)), one_gaussian((-1.0, 0.0, 0.0), opacity_logit=0.0, dc=(-1.0, -1.0, -1.0))
    both = {k: torch.cat([far[k], near[k]]) for k in near}  # index order = back to front: the sort must undo it
    out = so.render(both, c2w, fx, fy, cx, cy, W, H, sh_degree_to_use=0)
    c_near = max(so.SH_C0 * 1.0 + 0.5, 0.0)
    c_far = max(so.SH_C0 * -1.0 + 0.5, 0.0)
    a = 0.5  # sigmoid(0) at the centre of both (same optical axis)
    want = a * c_near + (1 - a) * a * c_far
    assert abs(float(out["rgb"][23, 31, 0]) - want) < 1e-5
    assert abs(float(out["accumulation"][23, 31]) - (1 - (1 - a) ** 2)) < 1e-6
    d = (a * 2.0 + (1 - a) * a * 4.0) / (1 - (1 - a) ** 2)  # depths 2 and 4 from the eye at x = 3
    assert abs(float(out["depth"][23, 31]) - d) < 1e-4


def test_transmittance_stop_and_alpha_cutoff():
    c2w, fx, fy, cx, cy, W, H = cam()
    # 6 nearly opaque Gaussians on the axis: alpha = 0.999 each -> T = 1e-3 after the first, 1e-6 <= 1e-4 after the second: the pixel stops
    # BEFORE the second one is blended
    gs = [one_gaussian((1.0 - 0.3 * k, 0.0, 0.0), opacity_logit=20.0, dc=(float(k), 0.0, 0.0)) for k in range(6)]
    p = {k: torch.cat([g[k] for g in gs]) for k in gs[0]}
    out = so.render(p, c2w, fx, fy, cx, cy, W, H, sh_degree_to_use=0)
    assert abs(float(out["accumulation"][23, 31]) - 0.999) < 1e-6
    assert abs(float(out["rgb"][23, 31, 0]) - 0.999 * 0.5) < 1e-5  # only Gaussian 0 (dc 0 -> colour 0.5)
    # an almost transparent Gaussian (alpha < 1/255 everywhere) leaves no trace
    faint = one_gaussian((0.0, 0.0, 0.0), opacity_logit=-6.0)
    out = so.render(faint, c2w, fx, fy, cx, cy, W, H, sh_degree_to_use=0)
    assert float(out["accumulation"].max()) == 0.0


def test_empty_view_returns_the_reference_fallback():
    c2w, fx, fy, cx, cy, W, H = cam()
    out = so.render(one_gaussian((10.0, 0.0, 0.0)), c2w, fx, fy, cx, cy, W, H)  # behind the camera
    assert float(out["accumulation"].abs().max()) == 0.0 and float(out["depth"].min()) == 10.0 and out["rgb"].shape == (H, W, 3)


def test_c_abi_exports_the_splat_entry_points():
    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import _lib

    lib = _lib.load()
    for name in ("tn_splat_workspace_bytes", "tn_splat_project", "tn_splat_bin", "tn_splat_raster"):
        assert hasattr(lib, name)
