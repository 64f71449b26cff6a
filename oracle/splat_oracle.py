"""CPU oracle for the N4 row: forward Gaussian-splat render with an RGB + thermal colour per Gaussian.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path.

**PARITY UNPINNED.**  The reference has no thermal-splatfacto; the nearest code is stock `SplatfactoModel.get_outputs`
(/root/reference/nerfstudio/models/splatfacto.py:659-822), whose arithmetic lives in the third-party CUDA package `gsplat>=0.1.6`
(/root/reference/pyproject.toml:66) -- not vendored, not installed here, and the reference holds no test vectors for it
(tests/test_train.py:29-30 blacklists the method).  This file restates the PUBLISHED algorithm of gsplat 0.1.x (the API the call sites
use: project_gaussians(means, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, H, W, block_width) -> xys, depths, radii,
conics, compensation, num_tiles_hit, cov3d; spherical_harmonics(degree, viewdirs, coeffs); rasterize_gaussians(xys, depths, radii, conics,
num_tiles_hit, colors, opacity, H, W, block_width, background, return_alpha)), i.e. EWA splatting with a 0.3-pixel screen-space blur,
3-sigma radii, 16x16 tiles, per-tile front-to-back alpha blending with alpha clamp 0.999, cut-off 1/255 and transmittance stop 1e-4
(Kerbl et al. 2023; Zwicker et al. 2001).  Anchors are the reference's own call sites (splatfacto.py:722-822) only.

The thermal channel is this framework's extension in the spirit of thermal-nerfacto's shared density: the SAME Gaussians (means, scales,
rotations, opacities) carry a second set of spherical-harmonic coefficients with one channel, rendered through the same rasteriser.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

BLOCK_WIDTH = 16  # splatfacto.py:738
CLIP_THRESH = 0.01  # gsplat project_gaussians default
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> Tensor:
    """splatfacto.py:82-100 (OpenGL-style perspective matrix)."""
    t = znear * math.tan(0.5 * fovy)
    b = -t
    r = znear * math.tan(0.5 * fovx)
    l = -r  # noqa: E741
    n, f = znear, zfar
    return torch.tensor([[2 * n / (r - l), 0.0, (r + l) / (r - l), 0.0], [0.0, 2 * n / (t - b), (t + b) / (t - b), 0.0],
                         [0.0, 0.0, (f + n) / (f - n), -1.0 * f * n / (f - n)], [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32)


def camera_matrices(c2w: Tensor, fx: float, fy: float, width: int, height: int) -> Tuple[Tensor, Tensor]:
    """splatfacto.py:700-720: world->camera matrix in gsplat's convention (y down, z forward) and the full projection matrix."""
    R = c2w[:3, :3].float()
    T = c2w[:3, 3:4].float()
    R = R @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    R_inv = R.T
    T_inv = -R_inv @ T
    viewmat = torch.eye(4)
    viewmat[:3, :3] = R_inv
    viewmat[:3, 3:4] = T_inv
    fovx = 2 * math.atan(width / (2 * fx))
    fovy = 2 * math.atan(height / (2 * fy))
    projmat = projection_matrix(0.001, 1000, fovx, fovy)
    return viewmat, projmat @ viewmat


def quat_to_rotmat(q: Tensor) -> Tensor:
    """(w, x, y, z), normalised inside (gsplat quat_to_rotmat)."""
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                        2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def project_gaussians(means: Tensor, scales: Tensor, glob_scale: float, quats: Tensor, viewmat: Tensor, projmat: Tensor, fx: float, fy: float,
                      cx: float, cy: float, H: int, W: int, block_width: int = BLOCK_WIDTH, clip_thresh: float = CLIP_THRESH) -> Dict[str, Tensor]:
    """gsplat 0.1.x project_gaussians forward.  scales are the exponentiated scales, quats normalised by the caller (splatfacto.py:739-753)."""
    N = means.shape[0]
    Rv, tv = viewmat[:3, :3], viewmat[:3, 3]
    p_view = means @ Rv.T + tv  # [N,3]
    visible = p_view[:, 2] > clip_thresh
    # 3D covariance  Sigma = (R S)(R S)^T
    M = quat_to_rotmat(quats) * (glob_scale * scales)[:, None, :]
    cov3d = M @ M.transpose(1, 2)
    # EWA projection: clamp the view-space position to 1.3x the frustum, J = perspective Jacobian, cov2d = (J W) Sigma (J W)^T + 0.3 I
    tan_fovx, tan_fovy = 0.5 * W / fx, 0.5 * H / fy
    lim_x, lim_y = 1.3 * tan_fovx, 1.3 * tan_fovy
    tz = p_view[:, 2]
    tx = tz * torch.clamp(p_view[:, 0] / tz, -lim_x, lim_x)
    ty = tz * torch.clamp(p_view[:, 1] / tz, -lim_y, lim_y)
    rz = 1.0 / tz
    rz2 = rz * rz
    J = torch.zeros(N, 2, 3)
    J[:, 0, 0] = fx * rz
    J[:, 0, 2] = -fx * tx * rz2
    J[:, 1, 1] = fy * rz
    J[:, 1, 2] = -fy * ty * rz2
    Tm = J @ Rv
    cov = Tm @ cov3d @ Tm.transpose(1, 2)
    a0, b0, c0 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det_orig = a0 * c0 - b0 * b0
    a, b, c = a0 + 0.3, b0, c0 + 0.3
    det = a * c - b * b
    comp = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    ok = visible & (det != 0)
    inv_det = 1.0 / det
    conics = torch.stack([c * inv_det, -b * inv_det, a * inv_det], -1)
    mid = 0.5 * (a + c)
    disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    v1, v2 = mid + disc, mid - disc
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(v1, v2)))
    # pixel centre: project with the full matrix, ndc -> pixel with the principal point (ndc2pix: 0.5 W x + cx - 0.5)
    ph = torch.cat([means, torch.ones(N, 1)], -1) @ projmat.T
    rw = 1.0 / (ph[:, 3] + 1e-6)
    xys = torch.stack([0.5 * W * (ph[:, 0] * rw) + cx - 0.5, 0.5 * H * (ph[:, 1] * rw) + cy - 0.5], -1)
    # tile bounding box
    tb_x, tb_y = (W + block_width - 1) // block_width, (H + block_width - 1) // block_width
    tcx, tcy, tr = xys[:, 0] / block_width, xys[:, 1] / block_width, radius / block_width
    clampi = lambda v, hi: torch.clamp(v.to(torch.int32), 0, hi)  # noqa: E731   ((int) truncation, then min(max(0, .), bound))
    safe = lambda v: torch.where(ok, v, torch.zeros_like(v))  # noqa: E731
    x0, x1 = clampi(safe(tcx - tr), tb_x), clampi(safe(tcx + tr + 1), tb_x)
    y0, y1 = clampi(safe(tcy - tr), tb_y), clampi(safe(tcy + tr + 1), tb_y)
    area = (x1 - x0) * (y1 - y0)
    ok = ok & (area > 0)
    z = lambda t: torch.where(ok.reshape(-1, *[1] * (t.dim() - 1)), t, torch.zeros_like(t))  # noqa: E731
    return {"xys": z(xys), "depths": z(tz), "radii": z(radius).to(torch.int32), "conics": z(conics), "compensation": z(comp),
            "num_tiles_hit": z(area).to(torch.int32), "tile_min": torch.stack([x0, y0], -1), "tile_max": torch.stack([x1, y1], -1), "cov3d": cov3d}


def spherical_harmonics(degree: int, dirs: Tensor, coeffs: Tensor) -> Tensor:
    """gsplat spherical_harmonics forward: dirs [N,3] (normalised inside for degree >= 1), coeffs [N,K,C] -> [N,C]."""
    c = coeffs
    out = SH_C0 * c[:, 0]
    if degree < 1:
        return out
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    out = out + SH_C1 * (-y * c[:, 1] + z * c[:, 2] - x * c[:, 3])
    if degree < 2:
        return out
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    out = out + (SH_C2[0] * xy * c[:, 4] + SH_C2[1] * yz * c[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * c[:, 6] + SH_C2[3] * xz * c[:, 7]
                 + SH_C2[4] * (xx - yy) * c[:, 8])
    if degree < 3:
        return out
    out = out + (SH_C3[0] * y * (3.0 * xx - yy) * c[:, 9] + SH_C3[1] * xy * z * c[:, 10] + SH_C3[2] * y * (4.0 * zz - xx - yy) * c[:, 11]
                 + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * c[:, 12] + SH_C3[4] * x * (4.0 * zz - xx - yy) * c[:, 13]
                 + SH_C3[5] * z * (xx - yy) * c[:, 14] + SH_C3[6] * x * (xx - 3.0 * yy) * c[:, 15])
    return out


def rasterize_gaussians(xys: Tensor, depths: Tensor, radii: Tensor, conics: Tensor, tile_min: Tensor, tile_max: Tensor, colors: Tensor,
                        opacity: Tensor, H: int, W: int, block_width: int, background: Tensor) -> Tuple[Tensor, Tensor]:
    """gsplat rasterize_gaussians forward (return_alpha=True): every pixel walks the Gaussians whose tile bounding box contains its tile in
    depth order (radix sort of (tile id, depth bits): stable, so equal depths keep the Gaussian order), pixel centre (j + 0.5, i + 0.5):
    sigma = 0.5 (cx dx^2 + cz dy^2) + cy dx dy; alpha = min(0.999, opacity exp(-sigma)); skipped when sigma < 0 or alpha < 1/255; the pixel
    stops BEFORE a Gaussian that would take its transmittance to <= 1e-4.  Vectorised over pixels, sequential over Gaussians."""
    C = colors.shape[1]
    order = torch.argsort(depths, stable=True)
    order = order[radii[order] > 0]
    py, px = torch.meshgrid(torch.arange(H, dtype=torch.float32) + 0.5, torch.arange(W, dtype=torch.float32) + 0.5, indexing="ij")
    T = torch.ones(H, W)
    done = torch.zeros(H, W, dtype=torch.bool)
    out = torch.zeros(H, W, C)
    for g in order.tolist():
        x0, y0 = (tile_min[g] * block_width).tolist()
        x1, y1 = (tile_max[g] * block_width).tolist()
        x1, y1 = min(x1, W), min(y1, H)
        if x1 <= x0 or y1 <= y0:
            continue
        sl = (slice(y0, y1), slice(x0, x1))
        dx = xys[g, 0] - px[sl]
        dy = xys[g, 1] - py[sl]
        sigma = 0.5 * (conics[g, 0] * dx * dx + conics[g, 2] * dy * dy) + conics[g, 1] * dx * dy
        alpha = torch.clamp(opacity[g] * torch.exp(-sigma), max=0.999)
        use = (sigma >= 0) & (alpha >= 1.0 / 255.0) & ~done[sl]
        next_T = T[sl] * (1.0 - alpha)
        stop = use & (next_T <= 1e-4)
        done[sl] |= stop
        use = use & ~stop
        vis = alpha * T[sl]
        out[sl] += torch.where(use[..., None], vis[..., None] * colors[g], torch.zeros(()))
        T[sl] = torch.where(use, next_T, T[sl])
    img = out + T[..., None] * background
    return img, 1.0 - T


def render(params: Dict[str, Tensor], c2w: Tensor, fx: float, fy: float, cx: float, cy: float, W: int, H: int, sh_degree_to_use: int = 3,
           rasterize_mode: str = "classic", background: Optional[Tensor] = None, background_thermal: float = 0.0) -> Dict[str, Tensor]:
    """SplatfactoModel.get_outputs (splatfacto.py:659-822), eval mode, no crop box, plus the thermal channel.
    params: means [N,3], scales [N,3] (log), quats [N,4], opacities [N,1] (logit), features_dc [N,3], features_rest [N,15,3],
    features_dc_thermal [N,1], features_rest_thermal [N,15,1]."""
    background = torch.zeros(3) if background is None else background
    viewmat, projmat = camera_matrices(c2w, fx, fy, W, H)
    means = params["means"].float()
    quats = params["quats"] / params["quats"].norm(dim=-1, keepdim=True)
    pj = project_gaussians(means, torch.exp(params["scales"]), 1.0, quats, viewmat, projmat, fx, fy, cx, cy, H, W)
    out: Dict[str, Tensor] = {"projection": pj, "background": background}
    if int(pj["radii"].sum()) == 0:  # splatfacto.py:759-764
        out.update(rgb=background.repeat(H, W, 1), thermal=torch.full((H, W, 1), background_thermal), depth=torch.full((H, W, 1), 10.0),
                   accumulation=torch.zeros(H, W, 1))
        return out
    viewdirs = means - c2w[:3, 3].float()
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    col = torch.cat([params["features_dc"][:, None, :], params["features_rest"]], 1)
    col_t = torch.cat([params["features_dc_thermal"][:, None, :], params["features_rest_thermal"]], 1)
    if sh_degree_to_use >= 0 and col.shape[1] > 1:
        rgbs = torch.clamp(spherical_harmonics(sh_degree_to_use, viewdirs, col) + 0.5, min=0.0)
        ths = torch.clamp(spherical_harmonics(sh_degree_to_use, viewdirs, col_t) + 0.5, min=0.0)
    else:  # sh_degree == 0 (splatfacto.py:776-777)
        rgbs, ths = torch.sigmoid(col[:, 0]), torch.sigmoid(col_t[:, 0])
    op_plain = torch.sigmoid(params["opacities"])[:, 0]
    if rasterize_mode == "antialiased":
        op = op_plain * pj["compensation"]
    elif rasterize_mode == "classic":
        op = op_plain
    else:
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    bg4 = torch.cat([background, torch.tensor([background_thermal])])
    img, alpha = rasterize_gaussians(pj["xys"], pj["depths"], pj["radii"], pj["conics"], pj["tile_min"], pj["tile_max"], torch.cat([rgbs, ths], -1),
                                     op, H, W, BLOCK_WIDTH, bg4)
    depth_im, _ = rasterize_gaussians(pj["xys"], pj["depths"], pj["radii"], pj["conics"], pj["tile_min"], pj["tile_max"], pj["depths"][:, None],
                                      op_plain, H, W, BLOCK_WIDTH, torch.zeros(1))
    alpha = alpha[..., None]
    depth = torch.where(alpha > 0, depth_im / alpha, depth_im.max())  # splatfacto.py:809
    out.update(rgb=torch.clamp(img[..., :3], max=1.0), thermal=torch.clamp(img[..., 3:], max=1.0), depth=depth, accumulation=alpha, colors=rgbs,
               colors_thermal=ths, opacities=op)
    return out


# synthetic scene / camera generators live in the product package (nerfstudio_thermal_amd/synth.py: bench inputs must not come from oracle/);
# re-exported here for the tests that build oracle and HIP inputs side by side
def synth_gaussians(*a, **k):
    from nerfstudio_thermal_amd.synth import synth_gaussians as f

    return f(*a, **k)


def look_at_camera(*a, **k):
    from nerfstudio_thermal_amd.synth import look_at_camera as f

    return f(*a, **k)
