"""CPU ORACLE for the thermal-nerfacto volume-rendering path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, array-in/array-out restatement (PyTorch-CPU, fp32) of
the reference's `implementation="torch"` path.  It is the checker that the HIP
kernels in `nerfstudio-thermal_amd/csrc` are compared against and the thing
`bench.py` times for its `cpu_baseline` leg.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it; the
product package never does.

Parity status: PINNED.  `oracle/make_golden.py` imports the reference itself in
the build container (via `oracle/ref_import.py`) and stores its outputs for
every function below under `tests/golden/`; `tests/test_oracle_vs_golden.py`
checks this file against those vectors (and against the live reference when
/root/reference is present).

Each function cites the reference file:line it restates (paths relative to
/root/reference/nerfstudio/).

Parameters travel as a flat dict {reference state_dict key: tensor}; see
`param_keys()`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

# --------------------------------------------------------------------------------------
# configuration (defaults = models/nerfacto.py:52-133 + models/thermal_nerfacto.py:32-64)
# --------------------------------------------------------------------------------------


@dataclass
class OracleConfig:
    density_mode: str = "shared"  # "shared" | "separate"
    num_images: int = 8
    # main field hash grid (fields/nerfacto_field.py:73-100)
    num_levels: int = 16
    base_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    geo_feat_dim: int = 15
    appearance_embed_dim: int = 32
    # proposal nets (models/nerfacto.py:91-96)
    prop_num_levels: int = 5
    prop_base_res: int = 16
    prop_max_res: Tuple[int, ...] = (128, 256)
    prop_log2_hashmap_size: int = 17
    prop_hidden_dim: int = 16
    # sampling (models/nerfacto.py:79-82)
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    near_plane: float = 0.05
    far_plane: float = 1000.0
    # loss multipliers (models/nerfacto.py:100-103, models/thermal_nerfacto.py:37-56)
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    thermal_loss_mult: float = 100.0
    tv_pixel_loss_mult: float = 1e-6
    cross_channel_loss_mult: float = 1e-6
    density_loss_mult: float = 5e-5
    rgb_density_loss_mult: float = 0.01
    removal_min_density_diff: float = 0.05
    # camera optimizer regulariser (cameras/camera_optimizers.py:47-56; thermal twin penalty_scale=10)
    trans_l2_penalty: float = 1e-2
    rot_l2_penalty: float = 1e-3
    penalty_scale: float = 1.0
    penalty_scale_thermal: float = 10.0
    is_thermal_cam: Tuple[int, ...] = (0, 0, 0, 0, 1, 1, 1, 1)

    @property
    def num_channels(self) -> int:  # models/thermal_nerfacto.py:110
        return 3 + (1 if self.density_mode == "shared" else 0)


def level_resolutions(num_levels: int, min_res: int, max_res: int) -> Tensor:
    """Per-level grid scale.  field_components/encodings.py:343-345.

    growth is a numpy float64, the power is evaluated by torch on an int64
    arange (-> float32), then floored: the top level of the default main grid
    comes out as 2047, not 2048.
    """
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * growth**levels).to(torch.float32)


# --------------------------------------------------------------------------------------
# parameter naming (the reference state_dict keys; SURVEY.md 8b tier 2)
# --------------------------------------------------------------------------------------


def field_keys(prefix: str) -> Dict[str, str]:
    return {
        "table": f"{prefix}.mlp_base.model.0.hash_table",
        "w0": f"{prefix}.mlp_base.model.1.layers.0.weight",
        "b0": f"{prefix}.mlp_base.model.1.layers.0.bias",
        "w1": f"{prefix}.mlp_base.model.1.layers.1.weight",
        "b1": f"{prefix}.mlp_base.model.1.layers.1.bias",
        "hw0": f"{prefix}.mlp_head.layers.0.weight",
        "hb0": f"{prefix}.mlp_head.layers.0.bias",
        "hw1": f"{prefix}.mlp_head.layers.1.weight",
        "hb1": f"{prefix}.mlp_head.layers.1.bias",
        "hw2": f"{prefix}.mlp_head.layers.2.weight",
        "hb2": f"{prefix}.mlp_head.layers.2.bias",
        "emb": f"{prefix}.embedding_appearance.embedding.weight",
    }


def prop_keys(prefix: str, i: int) -> Dict[str, str]:
    return {
        "table": f"{prefix}.{i}.mlp_base.0.hash_table",
        "table_alias": f"{prefix}.{i}.encoding.hash_table",
        "w0": f"{prefix}.{i}.mlp_base.1.layers.0.weight",
        "b0": f"{prefix}.{i}.mlp_base.1.layers.0.bias",
        "w1": f"{prefix}.{i}.mlp_base.1.layers.1.weight",
        "b1": f"{prefix}.{i}.mlp_base.1.layers.1.bias",
    }


def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """All trainable tensors of the model, keyed as the reference names them."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    F = cfg.features_per_level

    def add_field(prefix: str, channels: int):
        k = field_keys(prefix)
        T = 2**cfg.log2_hashmap_size
        shapes[k["table"]] = (T * cfg.num_levels, F)
        shapes[k["w0"]] = (cfg.hidden_dim, cfg.num_levels * F)
        shapes[k["b0"]] = (cfg.hidden_dim,)
        shapes[k["w1"]] = (1 + cfg.geo_feat_dim, cfg.hidden_dim)
        shapes[k["b1"]] = (1 + cfg.geo_feat_dim,)
        din = 16 + cfg.geo_feat_dim + cfg.appearance_embed_dim
        shapes[k["hw0"]] = (cfg.hidden_dim_color, din)
        shapes[k["hb0"]] = (cfg.hidden_dim_color,)
        shapes[k["hw1"]] = (cfg.hidden_dim_color, cfg.hidden_dim_color)
        shapes[k["hb1"]] = (cfg.hidden_dim_color,)
        shapes[k["hw2"]] = (channels, cfg.hidden_dim_color)
        shapes[k["hb2"]] = (channels,)
        shapes[k["emb"]] = (cfg.num_images, cfg.appearance_embed_dim)

    def add_props(prefix: str):
        for i in range(len(cfg.num_proposal_samples_per_ray)):
            k = prop_keys(prefix, i)
            T = 2**cfg.prop_log2_hashmap_size
            shapes[k["table"]] = (T * cfg.prop_num_levels, F)
            shapes[k["w0"]] = (cfg.prop_hidden_dim, cfg.prop_num_levels * F)
            shapes[k["b0"]] = (cfg.prop_hidden_dim,)
            shapes[k["w1"]] = (1, cfg.prop_hidden_dim)
            shapes[k["b1"]] = (1,)

    add_field("field", cfg.num_channels)
    add_props("proposal_networks")
    shapes["camera_optimizer.pose_adjustment"] = (cfg.num_images, 6)
    # constructed unconditionally by the reference (models/thermal_nerfacto.py:138-186)
    add_props("proposal_networks_thermal")
    shapes["camera_optimizer_thermal.pose_adjustment"] = (cfg.num_images, 6)
    if cfg.density_mode == "separate":
        add_field("field_thermal", 1)
    return shapes


# --------------------------------------------------------------------------------------
# element-wise pieces
# --------------------------------------------------------------------------------------


class _TruncExp(torch.autograd.Function):
    """exp() whose backward clamps the exponent to [-15, 15].  field_components/activations.py:28-41."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def contract_linf(x: Tensor) -> Tensor:
    """mip-NeRF-360 contraction with the L-infinity norm.  field_components/spatial_distortions.py:66-69."""
    mag = torch.linalg.norm(x, ord=float("inf"), dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def unit_cube_positions(positions: Tensor) -> Tuple[Tensor, Tensor]:
    """contraction -> [0,1]^3 -> selector mask.  fields/density_fields.py:96-103 == fields/nerfacto_field.py:207-215."""
    p = contract_linf(positions)
    p = (p + 2.0) / 4.0
    selector = ((p > 0.0) & (p < 1.0)).all(dim=-1)
    p = p * selector[..., None]
    return p, selector


_PRIMES = (1, 2654435761, 805459861)


def hash_index(ix: Tensor, iy: Tensor, iz: Tensor, table_size: int, level_offset: Tensor) -> Tensor:
    """field_components/encodings.py:401-418 (int32 coords * int64 primes, xor, mod 2^k, + level offset)."""
    h = (ix.to(torch.int64) * _PRIMES[0]) ^ (iy.to(torch.int64) * _PRIMES[1]) ^ (iz.to(torch.int64) * _PRIMES[2])
    return h % table_size + level_offset


def hash_encode(x: Tensor, table: Tensor, res: Tensor, log2_hashmap_size: int) -> Tensor:
    """Multiresolution hash encoding, torch fallback semantics.  field_components/encodings.py:420-461.

    x [P,3] in [0,1]; table [L*T, F]; res [L] float.  Corners are ceil/floor (not floor/floor+1) and
    every level is hashed.  Interpolation order: x, then y, then z.
    """
    L = res.shape[0]
    T = 2**log2_hashmap_size
    scaled = x[..., None, :] * res.view(-1, 1)  # [P, L, 3]
    c = torch.ceil(scaled).to(torch.int32)
    f = torch.floor(scaled).to(torch.int32)
    o = scaled - f
    lvl = torch.arange(L) * T
    cx, cy, cz = c[..., 0], c[..., 1], c[..., 2]
    fx, fy, fz = f[..., 0], f[..., 1], f[..., 2]
    g = lambda a, b, d: table[hash_index(a, b, d, T, lvl)]  # noqa: E731  [P, L, F]
    f0, f1, f2, f3 = g(cx, cy, cz), g(cx, fy, cz), g(fx, fy, cz), g(fx, cy, cz)
    f4, f5, f6, f7 = g(cx, cy, fz), g(cx, fy, fz), g(fx, fy, fz), g(fx, cy, fz)
    ox, oy, oz = o[..., 0:1], o[..., 1:2], o[..., 2:3]
    f03 = f0 * ox + f3 * (1 - ox)
    f12 = f1 * ox + f2 * (1 - ox)
    f56 = f5 * ox + f6 * (1 - ox)
    f47 = f4 * ox + f7 * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    enc = f0312 * oz + f4756 * (1 - oz)
    return enc.flatten(-2, -1)


def mlp(x: Tensor, weights: List[Tensor], biases: List[Tensor], out_sigmoid: bool = False) -> Tensor:
    """ReLU MLP with biases.  field_components/mlp.py:159-178."""
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = torch.nn.functional.linear(x, w, b)
        if i < n - 1:
            x = torch.relu(x)
    if out_sigmoid:
        x = torch.sigmoid(x)
    return x


def sh16(d: Tensor) -> Tensor:
    """Degree-4 real spherical harmonics evaluated directly on the given vector.  utils/math.py:29-78.

    The caller passes (dir+1)/2, un-remapped (fields/base_field.py:136-142, fields/nerfacto_field.py:280-282).
    """
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x**2, y**2, z**2
    c = torch.zeros((*d.shape[:-1], 16), dtype=d.dtype)
    c[..., 0] = 0.28209479177387814
    c[..., 1] = 0.4886025119029199 * y
    c[..., 2] = 0.4886025119029199 * z
    c[..., 3] = 0.4886025119029199 * x
    c[..., 4] = 1.0925484305920792 * x * y
    c[..., 5] = 1.0925484305920792 * y * z
    c[..., 6] = 0.9461746957575601 * zz - 0.31539156525251999
    c[..., 7] = 1.0925484305920792 * x * z
    c[..., 8] = 0.5462742152960396 * (xx - yy)
    c[..., 9] = 0.5900435899266435 * y * (3 * xx - yy)
    c[..., 10] = 2.890611442640554 * x * y * z
    c[..., 11] = 0.4570457994644658 * y * (5 * zz - 1)
    c[..., 12] = 0.3731763325901154 * z * (5 * zz - 3)
    c[..., 13] = 0.4570457994644658 * x * (5 * zz - 1)
    c[..., 14] = 1.445305721320277 * z * (xx - yy)
    c[..., 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return c


# --------------------------------------------------------------------------------------
# fields
# --------------------------------------------------------------------------------------


def prop_density(params: Dict[str, Tensor], prefix: str, i: int, cfg: OracleConfig, positions: Tensor) -> Tensor:
    """HashMLPDensityField.get_density via Field.density_fn.  fields/density_fields.py:95-118, fields/base_field.py:48-68.

    positions [N,S,3] world space -> density [N,S,1].
    """
    k = prop_keys(prefix, i)
    p, sel = unit_cube_positions(positions)
    res = level_resolutions(cfg.prop_num_levels, cfg.prop_base_res, cfg.prop_max_res[i])
    enc = hash_encode(p.view(-1, 3), params[k["table"]], res, cfg.prop_log2_hashmap_size)
    h = mlp(enc, [params[k["w0"]], params[k["w1"]]], [params[k["b0"]], params[k["b1"]]])
    h = h.view(*positions.shape[:-1], 1)
    return trunc_exp(h) * sel[..., None]


def field_density(params: Dict[str, Tensor], prefix: str, cfg: OracleConfig, positions: Tensor):
    """NerfactoField.get_density.  fields/nerfacto_field.py:205-229 (average_init_density is 1.0 on this path).

    Returns density [N,S,1], geo features [N,S,15], density_before_activation [N,S,1], encoding [N*S, 2L].
    """
    k = field_keys(prefix)
    p, sel = unit_cube_positions(positions)
    res = level_resolutions(cfg.num_levels, cfg.base_res, cfg.max_res)
    enc = hash_encode(p.view(-1, 3), params[k["table"]], res, cfg.log2_hashmap_size)
    h = mlp(enc, [params[k["w0"]], params[k["w1"]]], [params[k["b0"]], params[k["b1"]]])
    h = h.view(*positions.shape[:-1], 1 + cfg.geo_feat_dim)
    pre, geo = torch.split(h, [1, cfg.geo_feat_dim], dim=-1)
    density = trunc_exp(pre) * sel[..., None]
    return density, geo, pre, enc


def field_color(
    params: Dict[str, Tensor],
    prefix: str,
    cfg: OracleConfig,
    directions: Tensor,
    geo: Tensor,
    camera_indices: Tensor,
    training: bool,
) -> Tensor:
    """NerfactoField.get_outputs with ThermalNerfactoField's head.  fields/nerfacto_field.py:272-348,
    fields/thermal_nerfacto_field.py:91-99.

    directions [N,3] (per ray), geo [N,S,15], camera_indices [N] -> rgb(t) [N,S,C] after sigmoid.
    Train: per-camera appearance embedding; eval: mean embedding (use_average_appearance_embedding=True).
    """
    k = field_keys(prefix)
    N, S = geo.shape[:2]
    with torch.no_grad():  # field_components/encodings.py:792
        d = sh16((directions + 1.0) / 2.0)
    d = d[:, None, :].expand(N, S, 16)
    emb_w = params[k["emb"]]
    if training:
        e = emb_w[camera_indices][:, None, :].expand(N, S, cfg.appearance_embed_dim)
    else:
        e = torch.ones((N, S, cfg.appearance_embed_dim)) * emb_w.mean(dim=0)
    h = torch.cat([d.reshape(N * S, 16), geo.reshape(N * S, -1), e.reshape(N * S, -1)], dim=-1)
    rgb = mlp(
        h,
        [params[k["hw0"]], params[k["hw1"]], params[k["hw2"]]],
        [params[k["hb0"]], params[k["hb1"]], params[k["hb2"]]],
        out_sigmoid=True,
    )
    return rgb.view(N, S, -1)


# --------------------------------------------------------------------------------------
# samplers / weights
# --------------------------------------------------------------------------------------


def spacing_fn(x: Tensor) -> Tensor:
    """model_components/ray_samplers.py:244."""
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def spacing_fn_inv(x: Tensor) -> Tensor:
    """model_components/ray_samplers.py:245."""
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


def s_to_euclidean(bins: Tensor, nears: Tensor, fars: Tensor) -> Tensor:
    """spacing_to_euclidean_fn closure.  model_components/ray_samplers.py:113-118."""
    s_near, s_far = spacing_fn(nears), spacing_fn(fars)
    return spacing_fn_inv(bins * s_far + (1 - bins) * s_near)


def spaced_bins(num_rays: int, num_samples: int, jitter: Optional[Tensor]) -> Tensor:
    """Level-0 bins in s-space.  model_components/ray_samplers.py:100-111.

    jitter: None (eval) or [N,1] in [0,1) (train, single_jitter=True).  Returns [N, S+1] (eval: [1,S+1] expanded).
    """
    bins = torch.linspace(0.0, 1.0, num_samples + 1)[None, ...]
    if jitter is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * jitter
    return bins.expand(num_rays, num_samples + 1)


def get_weights(deltas: Tensor, densities: Tensor) -> Tensor:
    """RaySamples.get_weights.  cameras/rays.py:128-150.  deltas, densities [N,S,1]."""
    dd = deltas * densities
    alphas = 1 - torch.exp(-dd)
    trans = torch.cumsum(dd[..., :-1, :], dim=-2)
    trans = torch.cat([torch.zeros((*trans.shape[:1], 1, 1)), trans], dim=-2)
    trans = torch.exp(-trans)
    return torch.nan_to_num(alphas * trans)


def pdf_resample(
    existing_bins: Tensor, weights: Tensor, num_samples: int, jitter: Optional[Tensor], eps: float = 1e-5
) -> Tensor:
    """PDFSampler (include_original=False, histogram_padding=0.01).  model_components/ray_samplers.py:301-360.

    existing_bins [N, Sp+1] (s-space), weights [N, Sp, 1] (already annealed) -> new s-space bins [N, S+1], detached.
    jitter: None (eval: bin centres) or [N,1] uniform (train, single jitter).
    """
    num_bins = num_samples + 1
    w = weights[..., 0] + 0.01
    w_sum = torch.sum(w, dim=-1, keepdim=True)
    padding = torch.relu(eps - w_sum)
    w = w + padding / w.shape[-1]
    w_sum = w_sum + padding
    pdf = w / w_sum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins)
    if jitter is not None:
        u = u.expand(size=(*cdf.shape[:-1], num_bins)) + jitter / num_bins
    else:
        u = (u + 1.0 / (2 * num_bins)).expand(size=(*cdf.shape[:-1], num_bins))
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, side="right")
    hi = existing_bins.shape[-1] - 1
    below = torch.clamp(inds - 1, 0, hi)
    above = torch.clamp(inds, 0, hi)
    cdf0, cdf1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(existing_bins, -1, below), torch.gather(existing_bins, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf0) / (cdf1 - cdf0), 0), 0, 1)
    return (b0 + t * (b1 - b0)).detach()


@dataclass
class Samples:
    """One sampling level: the tensors RaySamples/Frustums carry (cameras/rays.py:251-295)."""

    s_bins: Tensor  # [N, S+1] normalised (spacing_starts / spacing_ends)
    e_bins: Tensor  # [N, S+1] euclidean (frustums.starts / ends)

    @property
    def starts(self):
        return self.e_bins[..., :-1, None]

    @property
    def ends(self):
        return self.e_bins[..., 1:, None]

    @property
    def deltas(self):
        return self.ends - self.starts

    def positions(self, origins: Tensor, directions: Tensor) -> Tensor:
        """Frustums.get_positions.  cameras/rays.py:49-58."""
        return origins[:, None, :] + directions[:, None, :] * (self.starts + self.ends) / 2


def proposal_sample(
    params: Dict[str, Tensor],
    cfg: OracleConfig,
    prefix: str,
    origins: Tensor,
    directions: Tensor,
    nears: Tensor,
    fars: Tensor,
    anneal: float = 1.0,
    jitters: Optional[List[Tensor]] = None,
    prop_requires_grad: bool = True,
):
    """ProposalNetworkSampler.generate_ray_samples.  model_components/ray_samplers.py:577-618.

    jitters: None (eval) or list of 3 [N,1] uniforms (level 0 spaced, level 1 pdf, level 2 pdf).
    Returns (final Samples, weights_list, samples_list) with the proposal levels only in the lists.
    """
    N = origins.shape[0]
    counts = list(cfg.num_proposal_samples_per_ray) + [cfg.num_nerf_samples_per_ray]
    weights_list, samples_list = [], []
    weights, samples = None, None
    for lvl, S in enumerate(counts):
        jit = None if jitters is None else jitters[lvl]
        if lvl == 0:
            s_bins = spaced_bins(N, S, jit)
        else:
            s_bins = pdf_resample(samples.s_bins, torch.pow(weights, anneal), S, jit)
        samples = Samples(s_bins=s_bins, e_bins=s_to_euclidean(s_bins, nears, fars))
        if lvl < len(counts) - 1:
            pos = samples.positions(origins, directions)
            if prop_requires_grad:
                density = prop_density(params, prefix, lvl, cfg, pos)
            else:
                with torch.no_grad():
                    density = prop_density(params, prefix, lvl, cfg, pos)
            weights = get_weights(samples.deltas, density)
            weights_list.append(weights)
            samples_list.append(samples)
    return samples, weights_list, samples_list


# --------------------------------------------------------------------------------------
# renderers
# --------------------------------------------------------------------------------------


def composite_rgb(rgb: Tensor, weights: Tensor, training: bool) -> Tensor:
    """RGBRenderer / RGBTRenderer with background_color='last_sample'.
    model_components/renderers.py:118-133,238-245 (== :292-307,418-425 for 4 channels)."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = torch.sum(weights * rgb, dim=-2)
    acc = torch.sum(weights, dim=-2)
    comp = comp + rgb[..., -1, :] * (1.0 - acc)
    if not training:
        comp = torch.clamp(comp, min=0.0, max=1.0)
    return comp


def accumulation(weights: Tensor) -> Tensor:
    """AccumulationRenderer.  model_components/renderers.py:509."""
    return torch.sum(weights, dim=-2)


def depth_median(weights: Tensor, samples: Samples) -> Tensor:
    """DepthRenderer('median').  model_components/renderers.py:547-557."""
    steps = (samples.starts + samples.ends) / 2
    cw = torch.cumsum(weights[..., 0], dim=-1)
    split = torch.ones((*weights.shape[:-2], 1)) * 0.5
    idx = torch.searchsorted(cw, split, side="left")
    idx = torch.clamp(idx, 0, steps.shape[-2] - 1)
    return torch.gather(steps[..., 0], dim=-1, index=idx)


def depth_expected(weights: Tensor, samples: Samples) -> Tensor:
    """DepthRenderer('expected'); clipped to the batch-global [min,max] of the sample midpoints.
    model_components/renderers.py:558-576."""
    steps = (samples.starts + samples.ends) / 2
    depth = torch.sum(weights * steps, dim=-2) / (torch.sum(weights, -2) + 1e-10)
    return torch.clip(depth, steps.min(), steps.max())


# --------------------------------------------------------------------------------------
# camera optimiser and ray generation
# --------------------------------------------------------------------------------------


def exp_map_so3xr3(tangent: Tensor) -> Tensor:
    """[R|t] from (t, so3).  cameras/lie_groups.py:24-58 (Rodrigues with theta^2 clamped at 1e-4)."""
    log_rot = tangent[:, 3:]
    nrms = (log_rot * log_rot).sum(1)
    ang = torch.clamp(nrms, 1e-4).sqrt()
    inv = 1.0 / ang
    fac1 = inv * ang.sin()
    fac2 = inv * inv * (1.0 - ang.cos())
    zero = torch.zeros_like(log_rot[:, 0])
    K = torch.stack(
        [
            torch.stack([zero, -log_rot[:, 2], log_rot[:, 1]], -1),
            torch.stack([log_rot[:, 2], zero, -log_rot[:, 0]], -1),
            torch.stack([-log_rot[:, 1], log_rot[:, 0], zero], -1),
        ],
        dim=1,
    )
    R = fac1[:, None, None] * K + fac2[:, None, None] * torch.bmm(K, K) + torch.eye(3)[None]
    return torch.cat([R, tangent[:, :3, None]], dim=-1)


def apply_pose_adjustment(
    pose_adjustment: Tensor, frozen_cam: Tensor, camera_indices: Tensor, origins: Tensor, directions: Tensor
) -> Tuple[Tensor, Tensor]:
    """CameraOptimizer(mode=SO3xR3).apply_to_raybundle with non-trainable cameras forced to identity.
    cameras/camera_optimizers.py:130-176.  frozen_cam: bool [C]."""
    m = exp_map_so3xr3(pose_adjustment[camera_indices, :])
    ident = torch.eye(4)[:3, :4]
    m = torch.where(frozen_cam[camera_indices][:, None, None], ident[None], m)
    o = origins + m[:, :3, 3]
    d = torch.bmm(m[:, :3, :3], directions[..., None]).squeeze(-1)
    return o, d


def camera_opt_regularizer(pose_adjustment: Tensor, cfg: OracleConfig, scale: float) -> Tensor:
    """cameras/camera_optimizers.py:189-195."""
    return (
        pose_adjustment[:, :3].norm(dim=-1).mean() * cfg.trans_l2_penalty
        + pose_adjustment[:, 3:].norm(dim=-1).mean() * cfg.rot_l2_penalty
    ) * scale


_NORM_EPS = float(np.finfo(float).eps * 4.0)  # cameras/camera_utils.py:28


def undistort_opencv(coords: Tensor, dist: Tensor, eps: float = 1e-3, iters: int = 10) -> Tensor:
    """Newton undistortion of (k1,k2,k3,k4,p1,p2).  cameras/camera_utils.py:343-446."""
    xd, yd = coords[..., 0], coords[..., 1]
    k1, k2, k3, k4, p1, p2 = (dist[..., i] for i in range(6))
    x, y = xd, yd
    for _ in range(iters):
        r = x * x + y * y
        d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
        fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
        fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
        d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
        d_x = 2.0 * x * d_r
        d_y = 2.0 * y * d_r
        fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
        fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
        fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
        fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
        den = fy_x * fx_y - fx_x * fy_y
        xn = fx * fy_y - fy * fx_y
        yn = fy * fx_x - fx * fy_x
        ok = torch.abs(den) > eps
        x = x + torch.where(ok, xn / den, torch.zeros_like(den))
        y = y + torch.where(ok, yn / den, torch.zeros_like(den))
    return torch.stack([x, y], dim=-1)


def sample_pixels(images: List[Tensor], is_thermal: Tensor, image_idx: Tensor, num_rays: int, u: Tensor, patch_size: int = 2):
    """PatchPixelSampler.sample on a jagged image list (no masks): data/pixel_samplers.py:296-337 (collate_image_dataset_batch_list),
    :389-441 (PatchPixelSampler.sample_method, else-branch), as VanillaDataManager.next_train draws it (base_datamanager.py:538-547).

    images: one [H_i, W_i, 3] tensor per batch position; is_thermal [num_images] (by batch position); image_idx [num_images] = the dataset
    (camera) index of each batch position; u [total_patches, 3] = the uniforms torch.rand would return, image after image.
    Every image gets (num_rays // num_images) // patch^2 patches, the last one the remainder; a patch is patch x patch adjacent pixels in
    row-major order.  Returns indices [N,3] int64 (camera,row,col), image [N,3], is_thermal [N]."""
    num_images = len(images)
    pp = patch_size * patch_size
    per = num_rays // num_images
    all_idx, all_img, pos = [], [], 0
    produced = 0
    for i in range(num_images):
        H, W, _ = images[i].shape
        n_i = per if i < num_images - 1 else num_rays - (num_images - 1) * produced
        sub = n_i // pp
        ui = u[pos:pos + sub]
        pos += sub
        ind = ui * torch.tensor([1, H - patch_size, W - patch_size])  # float32 x int64 -> float32
        ind = ind.view(sub, 1, 1, 3).broadcast_to(sub, patch_size, patch_size, 3).clone()
        yys, xxs = torch.meshgrid(torch.arange(patch_size), torch.arange(patch_size), indexing="ij")
        ind[:, ..., 1] += yys
        ind[:, ..., 2] += xxs
        ind = torch.floor(ind).long().flatten(0, 2)
        ind[:, 0] = i
        produced = ind.shape[0]
        all_idx.append(ind)
        all_img.append(images[i][ind[:, 1], ind[:, 2]])
    indices = torch.cat(all_idx, dim=0)
    c = indices[:, 0].clone()
    image = torch.cat(all_img, dim=0)
    assert image.shape[0] == num_rays
    indices[:, 0] = image_idx[c]
    return indices, image, is_thermal[c]


def generate_rays(
    ray_indices: Tensor, c2w: Tensor, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor, distortion: Optional[Tensor]
):
    """RayGenerator.forward -> Cameras._generate_rays_from_coords, PERSPECTIVE cameras with OPENCV distortion.
    model_components/ray_generators.py:40-55; cameras/cameras.py:598-655,781-786,886-909.

    ray_indices [N,3] int64 (camera,row,col); c2w [C,3,4]; fx,fy,cx,cy [C]; distortion [C,6] or None.
    Returns origins [N,3], directions [N,3], pixel_area [N,1], directions_norm [N,1].
    """
    c = ray_indices[:, 0]
    y = ray_indices[:, 1].to(torch.float32) + 0.5
    x = ray_indices[:, 2].to(torch.float32) + 0.5
    fx_, fy_, cx_, cy_ = fx[c], fy[c], cx[c], cy[c]
    coord = torch.stack([(x - cx_) / fx_, (y - cy_) / fy_], -1)
    coord_x = torch.stack([(x - cx_ + 1) / fx_, (y - cy_) / fy_], -1)
    coord_y = torch.stack([(x - cx_) / fx_, (y - cy_ + 1) / fy_], -1)
    stack = torch.stack([coord, coord_x, coord_y], dim=0)  # [3,N,2]
    if distortion is not None and (distortion != 0).any():
        stack = undistort_opencv(stack, distortion[c][None].expand(3, -1, -1))
    stack = stack * torch.tensor([1.0, -1.0])
    dirs = torch.cat([stack, -torch.ones_like(stack[..., :1])], dim=-1)  # [3,N,3]
    rot = c2w[c][:, :3, :3]
    dirs = torch.sum(dirs[..., None, :] * rot, dim=-1)
    norm = torch.maximum(torch.linalg.vector_norm(dirs, dim=-1, keepdims=True), torch.tensor([_NORM_EPS]).to(dirs))
    dirs = dirs / norm
    origins = c2w[c][:, :3, 3]
    d0 = dirs[0]
    dx = torch.sqrt(torch.sum((d0 - dirs[1]) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((d0 - dirs[2]) ** 2, dim=-1))
    return origins, d0, (dx * dy)[..., None], norm[0]


# --------------------------------------------------------------------------------------
# model forward (get_outputs)
# --------------------------------------------------------------------------------------


def _branch_outputs(
    params, cfg: OracleConfig, field_prefix: str, prop_prefix: str, origins, directions, camera_indices, nears, fars,
    training: bool, anneal: float, jitters, prop_requires_grad: bool,
):
    """proposal sampler + NerfactoModel._get_outputs for one spectrum branch.  models/nerfacto.py:299-353."""
    samples, weights_list, samples_list = proposal_sample(
        params, cfg, prop_prefix, origins, directions, nears, fars, anneal, jitters, prop_requires_grad
    )
    pos = samples.positions(origins, directions)
    density, geo, pre, _ = field_density(params, field_prefix, cfg, pos)
    rgb = field_color(params, field_prefix, cfg, directions, geo, camera_indices, training)
    weights = get_weights(samples.deltas, density)
    weights_list = weights_list + [weights]
    samples_list = samples_list + [samples]
    out = {
        "rgb": composite_rgb(rgb, weights, training),
        "accumulation": accumulation(weights),
        "expected_depth": depth_expected(weights, samples),
        "density": density,
        "density_before_activation": pre,  # not a reference output key; kept for the 1e-4 density check
        "field_rgb": rgb,  # ditto (per-sample colours)
    }
    with torch.no_grad():
        out["depth"] = depth_median(weights, samples)
    for i in range(len(cfg.num_proposal_samples_per_ray)):
        out[f"prop_depth_{i}"] = depth_median(weights_list[i], samples_list[i])
    out["weights_list"] = weights_list
    out["samples_list"] = samples_list
    return out, samples


def get_outputs(
    params: Dict[str, Tensor],
    cfg: OracleConfig,
    origins: Tensor,
    directions: Tensor,
    camera_indices: Tensor,
    training: bool,
    anneal: float = 1.0,
    jitters: Optional[List[Tensor]] = None,
    jitters_thermal: Optional[List[Tensor]] = None,
    prop_requires_grad: bool = True,
) -> Dict[str, object]:
    """Model.forward (collider) + ThermalNerfactoModel.get_outputs.
    models/base_model.py:132-143; model_components/scene_colliders.py:186-191; models/thermal_nerfacto.py:403-489.

    camera_indices [N] int64.  Train mode applies the SO3xR3 pose correction (RGB cameras trainable in the
    rgb optimiser, thermal cameras in the thermal twin).
    """
    N = origins.shape[0]
    near = cfg.near_plane if training else 0.0
    nears = torch.ones((N, 1)) * near
    fars = torch.ones((N, 1)) * cfg.far_plane
    is_th = torch.tensor(cfg.is_thermal_cam, dtype=torch.bool)
    o, d = origins, directions
    if training:
        o, d = apply_pose_adjustment(params["camera_optimizer.pose_adjustment"], is_th, camera_indices, o, d)
    out, samples = _branch_outputs(
        params, cfg, "field", "proposal_networks", o, d, camera_indices, nears, fars, training, anneal, jitters,
        prop_requires_grad,
    )
    if cfg.density_mode == "shared":
        rgbt = out["rgb"]
        out["rgbt"] = rgbt
        out["rgb"] = rgbt[..., :3]
        out["rgb_thermal"] = rgbt[..., 3:]
        return out
    # separate: second sampler + thermal field on a private copy of the bundle (models/thermal_nerfacto.py:430-445)
    ot, dt = origins, directions
    if training:
        ot, dt = apply_pose_adjustment(
            params["camera_optimizer_thermal.pose_adjustment"], ~is_th, camera_indices, ot, dt
        )
    # proposal_sampler_thermal never receives step_cb (use_proposal_thermal_weight_anneal=False): anneal stays 1.0
    # and its nets are always "updated" (models/thermal_nerfacto.py:222-250).
    out_t, samples_t = _branch_outputs(
        params, cfg, "field_thermal", "proposal_networks_thermal", ot, dt, camera_indices, nears, fars, training, 1.0,
        jitters_thermal, True,
    )
    for key, v in out_t.items():
        out[f"{key}_thermal"] = v
    # cross-evaluated densities (models/thermal_nerfacto.py:447-458)
    out["density2"] = field_density(params, "field", cfg, samples_t.positions(ot, dt))[0]
    out["density2_thermal"] = field_density(params, "field_thermal", cfg, samples.positions(o, d))[0]
    if not training:
        # removal renders (models/thermal_nerfacto.py:460-487); sigma/sigma is NaN where sigma == 0, as in the reference
        thr = cfg.removal_min_density_diff
        m = (out["density"] / out["density"] - out["density2_thermal"] / out["density"]).abs() < thr
        w = get_weights(samples.deltas, out["density"] * m)
        out["removal"] = composite_rgb(out["field_rgb"], w, training)
        m = (out["density_thermal"] / out["density_thermal"] - out["density2"] / out["density_thermal"]).abs() < thr
        # reference quirk: the thermal removal weights use the RGB branch's deltas (models/thermal_nerfacto.py:485)
        w = get_weights(samples.deltas, out["density_thermal"] * m)
        out["removal_thermal"] = composite_rgb(out["field_rgb_thermal"], w, training)
    return out


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------

_EPS = 1.0e-7  # model_components/losses.py:39


def _outer(t0s, t0e, t1s, t1e, y1):
    """model_components/losses.py:57-86."""
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    lo = torch.searchsorted(t1s.contiguous(), t0s.contiguous(), side="right") - 1
    lo = torch.clamp(lo, min=0, max=y1.shape[-1] - 1)
    hi = torch.searchsorted(t1e.contiguous(), t0e.contiguous(), side="right")
    hi = torch.clamp(hi, min=0, max=y1.shape[-1] - 1)
    return torch.take_along_dim(cy1[..., 1:], hi, dim=-1) - torch.take_along_dim(cy1[..., :-1], lo, dim=-1)


def interlevel_loss(weights_list: List[Tensor], samples_list: List[Samples]) -> Tensor:
    """model_components/losses.py:89-135."""
    c = samples_list[-1].s_bins.detach()
    w = weights_list[-1][..., 0].detach()
    total = 0.0
    for smp, wp in zip(samples_list[:-1], weights_list[:-1]):
        cp = smp.s_bins
        w_outer = _outer(c[..., :-1], c[..., 1:], cp[..., :-1], cp[..., 1:], wp[..., 0])
        total = total + torch.mean(torch.clip(w - w_outer, min=0) ** 2 / (w + _EPS))
    return total


def distortion_loss(weights_list: List[Tensor], samples_list: List[Samples]) -> Tensor:
    """model_components/losses.py:139-158."""
    t = samples_list[-1].s_bins
    w = weights_list[-1][..., 0]
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    intra = torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return torch.mean(inter + intra)


def rgb_to_rgbt(image: Tensor, is_thermal: Tensor) -> Tensor:
    """utils/rgbt_utils.py:6-32 (thermal pixels arrive as grey x3)."""
    rgbt = torch.zeros(image.shape[:-1] + (4,))
    rgbt[..., :3] = image * (1 - is_thermal)[:, None]
    rgbt[..., 3] = image[..., 0] * is_thermal
    return rgbt


def tv_pixel_loss(pred_thermal: Tensor, is_thermal: Tensor) -> Tensor:
    """model_components/losses.py:602-620 (2x2 patches of the RGB-camera rays)."""
    p = pred_thermal[(1 - is_thermal).bool()].view(-1, 4)
    return 0.25 * torch.mean(
        (p[:, 0] - p[:, 1]).abs() + (p[:, 0] - p[:, 2]).abs() + (p[:, 1] - p[:, 3]).abs() + (p[:, 2] - p[:, 3]).abs()
    )


def _pixel_grad(img: Tensor) -> Tensor:
    """model_components/losses.py:623-634."""
    p = img.view(-1, 4)
    return torch.stack((p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 1], p[:, 3] - p[:, 2]))


def cross_channel_loss(pred_thermal: Tensor, gt_rgb: Tensor, is_thermal: Tensor) -> Tensor:
    """model_components/losses.py:637-651."""
    keep = (1 - is_thermal).bool()
    diff = (_pixel_grad(pred_thermal[keep]) - _pixel_grad(gt_rgb[keep].mean(-1, keepdim=True))).abs()
    return 0.25 * (diff[0] + diff[1] + diff[2] + diff[3]).mean()


def loss_dict(
    params: Dict[str, Tensor], cfg: OracleConfig, outputs: Dict[str, object], image: Tensor, is_thermal: Tensor,
    training: bool = True,
) -> Dict[str, Tensor]:
    """ThermalNerfactoModel.get_metrics_dict['distortion'] + get_loss_dict.  models/thermal_nerfacto.py:270-388.

    image [N,3] float; is_thermal [N] float (0/1).
    """
    mse = torch.nn.functional.mse_loss
    pred = torch.cat((outputs["rgb"], outputs["rgb_thermal"]), dim=1)
    gt = rgb_to_rgbt(image, is_thermal)  # background 'last_sample' -> no blending (renderers.py:385-392)
    rgb_m, th_m = (1 - is_thermal)[:, None], is_thermal[:, None]
    out = {
        "rgb_loss": mse(pred[..., :3] * rgb_m, gt[..., :3] * rgb_m),
        "thermal_loss": cfg.thermal_loss_mult * mse(pred[..., 3:] * th_m, gt[..., 3:] * th_m),
    }
    if cfg.density_mode == "separate" and cfg.density_loss_mult > 0:
        l1 = torch.nn.functional.l1_loss
        a, b = cfg.density_loss_mult, cfg.rgb_density_loss_mult * cfg.density_loss_mult
        out["density_loss"] = (
            a * l1(outputs["density2"].detach(), outputs["density_thermal"])
            + a * l1(outputs["density"].detach(), outputs["density2_thermal"])
            + b * l1(outputs["density2"], outputs["density_thermal"].detach())
            + b * l1(outputs["density"], outputs["density2_thermal"].detach())
        )
    out["tv_pixel_loss"] = cfg.tv_pixel_loss_mult * tv_pixel_loss(pred[..., 3:], is_thermal)
    out["cross_channel_loss"] = cfg.cross_channel_loss_mult * cross_channel_loss(pred[..., 3:], gt[..., :3], is_thermal)
    if training:
        suffixes = ("", "_thermal") if cfg.density_mode == "separate" else ("",)
        out["interlevel_loss"] = 0
        out["distortion_loss"] = 0
        dist = 0
        for s in suffixes:
            dist = dist + distortion_loss(outputs[f"weights_list{s}"], outputs[f"samples_list{s}"])
        for s in suffixes:  # NB: the summed distortion is added once per suffix (models/thermal_nerfacto.py:363-368)
            out["interlevel_loss"] = out["interlevel_loss"] + cfg.interlevel_loss_mult * interlevel_loss(
                outputs[f"weights_list{s}"], outputs[f"samples_list{s}"]
            )
            out["distortion_loss"] = out["distortion_loss"] + cfg.distortion_loss_mult * dist
        out["camera_opt_regularizer"] = camera_opt_regularizer(
            params["camera_optimizer.pose_adjustment"], cfg, cfg.penalty_scale
        )
        if cfg.density_mode == "separate":
            out["camera_opt_regularizer_thermal"] = camera_opt_regularizer(
                params["camera_optimizer_thermal.pose_adjustment"], cfg, cfg.penalty_scale_thermal
            )
    return out


# --------------------------------------------------------------------------------------
# optimiser (engine/optimizers.py:73-210 -> torch.optim.Adam(lr, eps=1e-15), no weight decay)
# --------------------------------------------------------------------------------------


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, eps: float = 1e-15,
              beta1: float = 0.9, beta2: float = 0.999) -> None:
    """In-place Adam update with torch.optim.Adam's (non-fused, non-amsgrad) arithmetic; `step` is 1-based."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def optimizer_groups(cfg: OracleConfig) -> Dict[str, Tuple[List[str], float]]:
    """Param-group -> (keys, lr).  models/nerfacto.py:256-261, models/thermal_nerfacto.py:390-401,
    configs/method_configs.py:274-307."""
    shapes = param_shapes(cfg)
    groups = {
        "proposal_networks": ([k for k in shapes if k.startswith("proposal_networks.")], 1e-2),
        "fields": ([k for k in shapes if k.startswith("field.")], 1e-2),
        "camera_opt": (["camera_optimizer.pose_adjustment"], 1e-3),
    }
    if cfg.density_mode == "separate":
        groups["proposal_networks_thermal"] = ([k for k in shapes if k.startswith("proposal_networks_thermal.")], 1e-2)
        groups["fields_thermal"] = ([k for k in shapes if k.startswith("field_thermal.")], 1e-2)
        groups["camera_opt_thermal"] = (["camera_optimizer_thermal.pose_adjustment"], 1e-3)
    return groups


def exp_decay_lr(step: int, lr_init: float, lr_final: float, max_steps: int) -> float:
    """ExponentialDecayScheduler without warm-up.  engine/schedulers.py:109-141."""
    t = float(np.clip(step / max_steps, 0, 1))
    return float(np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))
