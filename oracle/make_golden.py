"""Generate golden vectors from the REFERENCE itself (build container only; test infrastructure).

    python oracle/make_golden.py            # writes tests/golden/*.npz (the "tiny" set: 2^12 / 2^10 tables, 32 rays)
    python oracle/make_golden.py --default --rays 256   # the same at 256 rays -> model_{shared,separate}_default256.npz
    python oracle/make_golden.py --default  # writes tests/golden/model_{shared,separate}_default.npz: the reference at its DEFAULT table
                                            # sizes (16 x 2^19 main, 5 x 2^17 proposal) on 64 rays; only the rays and the outputs are stored,
                                            # the 22 M / 39 M parameters are regenerated on both sides by synth.synth_params

Imports /root/reference through oracle/ref_import.py, builds the reference's own
ThermalNerfactoModel (implementation="torch") at a small table size, loads the
deterministic synthetic weights of nerfstudio-thermal_amd/synth.py into it, runs the
reference's own functions on synthetic rays and stores INPUT SEEDS + OUTPUTS.
Only arrays are stored: no reference source, bytecode or pickled modules.

Training-mode randomness (torch.rand in model_components/ray_samplers.py:105,322) is
replaced by injected jitter tensors for the duration of the call, so that the
train-mode goldens are reproducible by the oracle and by the HIP path.
"""
import os
import sys
import warnings
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import ref_import  # noqa: E402

ref_import.import_reference()

import nerfstudio_thermal_amd  # noqa: E402,F401  (alias loader for the hyphenated package dir)
from nerfstudio_thermal_amd import synth  # noqa: E402
import thermal_nerfacto_oracle as orc  # noqa: E402

from nerfstudio.cameras.cameras import Cameras, CameraType  # noqa: E402
from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.data.scene_box import SceneBox  # noqa: E402
from nerfstudio.model_components.ray_generators import RayGenerator  # noqa: E402
from nerfstudio.models.thermal_nerfacto import ThermalNerfactoModelConfig  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DEFAULT = "--default" in sys.argv
TINY = {} if DEFAULT else dict(log2_hashmap_size=12, prop_log2_hashmap_size=10)  # {} = the reference's defaults
# --default [--rays 256]: 64 rays by default; any other count goes to its own files (model_{mode}_default256.npz)
_RAYS = int(sys.argv[sys.argv.index("--rays") + 1]) if "--rays" in sys.argv else 64
N_RAYS = _RAYS if DEFAULT else 32
SUFFIX = ("_default" if _RAYS == 64 else f"_default{_RAYS}") if DEFAULT else ""
SEED = 0


def tiny_cfg(mode: str) -> orc.OracleConfig:
    return orc.OracleConfig(density_mode=mode, **TINY)


def build_reference_model(mode: str):
    cfg = tiny_cfg(mode)
    rc = ThermalNerfactoModelConfig(
        density_mode=mode,
        implementation="torch",
        log2_hashmap_size=cfg.log2_hashmap_size,
        proposal_net_args_list=[
            {"hidden_dim": 16, "log2_hashmap_size": cfg.prop_log2_hashmap_size, "num_levels": 5, "max_res": 128, "use_linear": False},
            {"hidden_dim": 16, "log2_hashmap_size": cfg.prop_log2_hashmap_size, "num_levels": 5, "max_res": 256, "use_linear": False},
        ],
    )
    model = rc.setup(
        scene_box=SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]])),
        num_train_data=cfg.num_images,
        metadata={"is_thermal": list(cfg.is_thermal_cam)},
    )
    shapes = orc.param_shapes(cfg)
    params = {k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, seed=SEED).items()}
    sd = model.state_dict()
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    # the aliased proposal tables appear under two keys
    full = dict(params)
    for pre in ("proposal_networks", "proposal_networks_thermal"):
        for i in range(2):
            pk = orc.prop_keys(pre, i)
            full[pk["table_alias"]] = params[pk["table"]]
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert not unexpected, unexpected
    assert all(("aabb" in m or "max_res" in m or "num_levels" in m or "log2_hashmap_size" in m or "device_indicator" in m)
               for m in missing), missing
    return model, cfg, params


class _InjectRand:
    """Replace torch.rand by a queue of given tensors (shape-checked)."""

    def __init__(self, queue):
        self.queue = list(queue)
        self.orig = torch.rand

    def __enter__(self):
        def fake(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list)):
                size = tuple(size[0])
            t = self.queue.pop(0)
            assert tuple(t.shape) == tuple(size), (t.shape, size)
            return t.clone()

        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self.orig
        assert not self.queue, "unused jitter"


def reference_cameras():
    cams = synth.synth_cameras()
    C = cams["c2w"].shape[0]
    ref = Cameras(
        camera_to_worlds=torch.from_numpy(cams["c2w"]),
        fx=torch.from_numpy(cams["fx"]),
        fy=torch.from_numpy(cams["fy"]),
        cx=torch.from_numpy(cams["cx"]),
        cy=torch.from_numpy(cams["cy"]),
        width=torch.from_numpy(cams["width"]),
        height=torch.from_numpy(cams["height"]),
        distortion_params=torch.from_numpy(cams["distortion"]),
        camera_type=CameraType.PERSPECTIVE,
    )
    assert ref.shape == (C,)
    return cams, ref


def tonp(x):
    return x.detach().cpu().numpy()


def golden_raygen():
    cams, ref = reference_cameras()
    idx = synth.synth_ray_indices(cams, N_RAYS)
    gen = RayGenerator(ref)
    rb = gen(torch.from_numpy(idx))
    np.savez_compressed(
        os.path.join(GOLDEN, "raygen.npz"),
        num_rays=N_RAYS,
        origins=tonp(rb.origins), directions=tonp(rb.directions), pixel_area=tonp(rb.pixel_area),
        camera_indices=tonp(rb.camera_indices), directions_norm=tonp(rb.metadata["directions_norm"]),
    )
    return rb


def golden_units():
    """Unit-level vectors: level scalings, hash encoding, SH, contraction, pose map, samplers."""
    from nerfstudio.cameras.lie_groups import exp_map_SO3xR3
    from nerfstudio.field_components.encodings import HashEncoding, SHEncoding
    from nerfstudio.field_components.spatial_distortions import SceneContraction
    from nerfstudio.model_components.ray_samplers import PDFSampler, UniformLinDispPiecewiseSampler
    from nerfstudio.model_components.scene_colliders import NearFarCollider

    out = {}
    out["res_main"] = tonp(HashEncoding(num_levels=16, min_res=16, max_res=2048, log2_hashmap_size=4, implementation="torch").scalings)
    out["res_prop0"] = tonp(HashEncoding(num_levels=5, min_res=16, max_res=128, log2_hashmap_size=4, implementation="torch").scalings)
    out["res_prop1"] = tonp(HashEncoding(num_levels=5, min_res=16, max_res=256, log2_hashmap_size=4, implementation="torch").scalings)

    # hash encoding on points that include exact lattice points and the unit-cube corners
    enc = HashEncoding(num_levels=16, min_res=16, max_res=2048, log2_hashmap_size=12, implementation="torch")
    table = torch.from_numpy(synth.uniform("unit_table", (16 * 4096, 2), seed=SEED) * np.float32(0.5))
    enc.hash_table = torch.nn.Parameter(table)
    x = torch.from_numpy(synth.uniform("unit_x", (256, 3), 0.0, 1.0, seed=SEED))
    x[0] = torch.tensor([0.0, 0.0, 0.0])
    x[1] = torch.tensor([0.5, 0.25, 0.125])
    x[2] = torch.tensor([1.0 - 2**-20, 0.0, 0.5])
    out["hash_enc"] = tonp(enc(x))

    d = torch.from_numpy(synth.uniform("unit_d", (64, 3), -1.0, 1.0, seed=SEED))
    d = d / d.norm(dim=-1, keepdim=True)
    out["sh16"] = tonp(SHEncoding(levels=4, implementation="torch")((d + 1.0) / 2.0))

    p = torch.from_numpy(synth.uniform("unit_p", (128, 3), -4.0, 4.0, seed=SEED))
    p[0] = torch.tensor([1.0, 0.2, -0.3])
    p[1] = torch.tensor([0.0, 0.0, 0.0])
    out["contract"] = tonp(SceneContraction(order=float("inf"))(p))

    tv = torch.from_numpy(synth.uniform("unit_pose", (8, 6), -0.2, 0.2, seed=SEED))
    tv[0, 3:] = 0.0  # exercises the theta^2 clamp
    tv[1, 3:] = torch.tensor([1e-3, -2e-3, 5e-4])
    out["exp_map"] = tonp(exp_map_SO3xR3(tv))

    # samplers: level-0 bins (eval/train) and one PDF resample (eval/train) on synthetic weights
    N = 16
    rays = synth.synth_rays_simple(N)
    rb = RayBundle(origins=torch.from_numpy(rays["origins"]), directions=torch.from_numpy(rays["directions"]),
                   pixel_area=torch.ones(N, 1), camera_indices=torch.from_numpy(rays["camera_indices"])[:, None])
    j0, j1, _ = (torch.from_numpy(j) for j in synth.synth_jitters(N))
    w = torch.from_numpy(synth.uniform("unit_w", (N, 256, 1), 0.0, 1.0, seed=SEED)) ** 8  # peaky
    w[3] = 0.0  # all-zero weights row -> padding guard
    for train in (False, True):
        col = NearFarCollider(near_plane=0.05, far_plane=1000.0)
        col.train(train)
        rbc = col(rb[...])
        s0 = UniformLinDispPiecewiseSampler(single_jitter=True)
        s0.train(train)
        pdf = PDFSampler(include_original=False, single_jitter=True)
        pdf.train(train)
        with _InjectRand([j0, j1] if train else []):
            rs0 = s0(rbc, num_samples=256)
            rs1 = pdf(rbc, rs0, w, num_samples=96)
        tag = "train" if train else "eval"
        out[f"spaced_s_{tag}"] = tonp(torch.cat([rs0.spacing_starts[..., 0], rs0.spacing_ends[..., -1:, 0]], -1))
        out[f"spaced_e_{tag}"] = tonp(torch.cat([rs0.frustums.starts[..., 0], rs0.frustums.ends[..., -1:, 0]], -1))
        out[f"pdf_s_{tag}"] = tonp(torch.cat([rs1.spacing_starts[..., 0], rs1.spacing_ends[..., -1:, 0]], -1))
        out[f"pdf_e_{tag}"] = tonp(torch.cat([rs1.frustums.starts[..., 0], rs1.frustums.ends[..., -1:, 0]], -1))
        if not train:
            dens = torch.from_numpy(synth.uniform("unit_dens", (N, 256, 1), 0.0, 40.0, seed=SEED))
            out["weights_from_density"] = tonp(rs0.get_weights(dens))
    np.savez_compressed(os.path.join(GOLDEN, "units.npz"), **out)


def sdist(rs):
    return torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], -1)


def edist(rs):
    return torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[..., -1:, 0]], -1)


def sample_indices(name, numel, k=2048):
    if numel <= k:
        return np.arange(numel, dtype=np.int64)
    return (synth.splitmix64(np.arange(k, dtype=np.uint64) + np.uint64(zlib.crc32(name.encode()))) % np.uint64(numel)).astype(np.int64)


def golden_model(mode: str, rb_src):
    model, cfg, params = build_reference_model(mode)
    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, N_RAYS)
    img, is_th = synth.synth_gt(idx, cams)
    jit = [torch.from_numpy(j) for j in synth.synth_jitters(N_RAYS)]
    jit_t = [torch.from_numpy(j) for j in synth.synth_jitters(N_RAYS, tag="_thermal")]
    out = {"mode": mode, "num_rays": N_RAYS}
    if DEFAULT:  # the tiny set reads its rays from raygen.npz
        out["rays/origins"], out["rays/directions"] = tonp(rb_src.origins), tonp(rb_src.directions)
        out["rays/camera_indices"] = tonp(rb_src.camera_indices)

    def bundle():
        return RayBundle(origins=rb_src.origins.clone(), directions=rb_src.directions.clone(),
                         pixel_area=rb_src.pixel_area.clone(), camera_indices=rb_src.camera_indices.clone())

    # ---------------- eval ----------------
    model.eval()
    with torch.no_grad():
        o = model(bundle())
    for k, v in o.items():
        if isinstance(v, torch.Tensor):
            out[f"eval/{k}"] = tonp(v)

    # ---------------- train (step 0: anneal as set_anneal(0) leaves it = 0.0? no: callbacks not run -> _anneal=1.0) ----
    model.train()
    # reproduce what the first training iteration does: set_anneal(step) is invoked by the trainer callback
    # (models/nerfacto.py:271-281); at step 0 the anneal is bias(0,10)=0, a degenerate all-equal weighting, so the
    # golden uses the step-500 value to exercise pow() with a non-trivial exponent.
    step = 500
    train_frac = np.clip(step / 1000, 0, 1)
    anneal = 10.0 * train_frac / ((10.0 - 1) * train_frac + 1)
    model.proposal_sampler.set_anneal(anneal)
    out["train/anneal"] = np.float64(anneal)
    queue = list(jit) + (list(jit_t) if mode == "separate" else [])
    with _InjectRand(queue):
        o = model(bundle())
    batch = {"image": torch.from_numpy(img), "is_thermal": torch.from_numpy(is_th)}
    metrics = {"distortion": 0}
    from nerfstudio.model_components.losses import distortion_loss

    for s in model.output_suffixes:
        metrics["distortion"] = metrics["distortion"] + distortion_loss(o[f"weights_list{s}"], o[f"ray_samples_list{s}"])
    losses = model.get_loss_dict(o, batch, metrics)
    total = sum(losses.values())
    total.backward()
    for k, v in o.items():
        if isinstance(v, torch.Tensor):
            out[f"train/{k}"] = tonp(v)
    for s in model.output_suffixes:
        for i, (w, rs) in enumerate(zip(o[f"weights_list{s}"], o[f"ray_samples_list{s}"])):
            out[f"train/weights{s}_{i}"] = tonp(w[..., 0])
            out[f"train/sbins{s}_{i}"] = tonp(sdist(rs))
            out[f"train/ebins{s}_{i}"] = tonp(edist(rs))
    for k, v in losses.items():
        out[f"loss/{k}"] = tonp(v) if isinstance(v, torch.Tensor) else np.float32(v)
    out["loss/total"] = tonp(total)
    sd = dict(model.named_parameters(remove_duplicate=False))
    for k in params:
        p = sd[k]
        g = p.grad
        if g is None:
            out[f"grad_none/{k}"] = np.int8(1)
            continue
        g = g.reshape(-1)
        out[f"grad_norm/{k}"] = np.float64(g.double().norm().item())
        ii = sample_indices(k, g.numel())
        out[f"grad_idx/{k}"] = ii
        out[f"grad_val/{k}"] = tonp(g[torch.from_numpy(ii)])

    # ---------------- one Adam step on the reference parameters (torch.optim.Adam as engine/optimizers.py builds it) ----
    groups = orc.optimizer_groups(cfg)
    for gname, (keys, lr) in groups.items():
        ps = [sd[k] for k in keys if sd[k].grad is not None]
        opt = torch.optim.Adam(ps, lr=lr, eps=1e-15)
        opt.step()
    for k in params:
        p = sd[k].detach().reshape(-1)
        ii = sample_indices(k, p.numel())
        out[f"adam_val/{k}"] = tonp(p[torch.from_numpy(ii)])
    np.savez_compressed(os.path.join(GOLDEN, f"model_{mode}{SUFFIX}.npz"), **out)
    # the checkpoint contract: every state_dict key of the reference model with its shape and dtype (default table sizes differ only in shape)
    import json

    if DEFAULT:
        print(mode, "default sizes", {k: float(v) for k, v in losses.items()})
        return
    with open(os.path.join(GOLDEN, f"state_dict_keys_{mode}.json"), "w") as f:
        json.dump({k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}, f, indent=0, sort_keys=True)
    print(mode, {k: float(v) for k, v in losses.items()})


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    if DEFAULT:
        cams, ref = reference_cameras()
        rb = RayGenerator(ref)(torch.from_numpy(synth.synth_ray_indices(cams, N_RAYS)))
        for mode in ("shared", "separate"):
            golden_model(mode, rb)
        return
    rb = golden_raygen()
    golden_units()
    for mode in ("shared", "separate"):
        golden_model(mode, rb)
    for f in sorted(os.listdir(GOLDEN)):
        print(f, os.path.getsize(os.path.join(GOLDEN, f)))


if __name__ == "__main__":
    main()
