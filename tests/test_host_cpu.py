"""Host-side logic that needs no GPU: the configuration mirror (defaults = the reference's, unsupported settings rejected loudly), the parameter
arena (layout, views, state-dict contract), ray containers, LR schedule, the method plugin object."""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

import nerfstudio_thermal_amd  # noqa: F401
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.config import CameraOptimizerConfig, ThermalNerfactoModelConfig
from nerfstudio_thermal_amd.engine import OPTIMIZERS, exp_decay_lr
from nerfstudio_thermal_amd.rays import Frustums, RayBundle

sys_path_oracle = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_config_defaults_equal_the_reference_config():
    """Field by field against the reference's own dataclass when it is importable here (the build container), otherwise against the values
    SURVEY.md 8d lists."""
    c = ThermalNerfactoModelConfig()
    assert (c.num_levels, c.base_res, c.max_res, c.log2_hashmap_size, c.features_per_level) == (16, 16, 2048, 19, 2)
    assert c.num_proposal_samples_per_ray == (256, 96) and c.num_nerf_samples_per_ray == 48 and (c.near_plane, c.far_plane) == (0.05, 1000.0)
    assert [a["max_res"] for a in c.proposal_net_args_list] == [128, 256] and all(a["log2_hashmap_size"] == 17 for a in c.proposal_net_args_list)
    import sys

    sys.path.insert(0, sys_path_oracle)
    import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference not present: defaults checked against SURVEY values only")
    ref_import.install_stubs()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    from nerfstudio.models.thermal_nerfacto import ThermalNerfactoModelConfig as RefCfg

    ref = RefCfg()
    mine = {f.name for f in dataclasses.fields(c)}
    skipped = []
    for f in dataclasses.fields(ref):
        if f.name == "_target":
            continue
        assert f.name in mine, f"missing config field {f.name}"
        a, b = getattr(c, f.name), getattr(ref, f.name)
        if dataclasses.is_dataclass(b):  # camera optimizer configs: compare the plain fields both sides have
            for g in dataclasses.fields(b):
                if g.name in ("_target", "optimizer", "scheduler"):
                    continue
                assert getattr(a, g.name) == getattr(b, g.name), (f.name, g.name)
            continue
        if f.name == "implementation":
            skipped.append(f.name)  # "tcnn"/"torch" there, the HIP library here
            continue
        assert a == b or (isinstance(a, (tuple, list)) and list(a) == list(b)), (f.name, a, b)
    assert skipped == ["implementation"] or skipped == []


@pytest.mark.parametrize("kw", [
    {"density_mode": "rgb_only"}, {"predict_normals": True}, {"num_proposal_samples_per_ray": (512, 96)}, {"num_proposal_iterations": 3},
    {"background_color": "random"}, {"use_same_proposal_network": True}, {"num_levels": 8}, {"tv_rgb_loss_mult": 1.0},
    {"camera_optimizer": CameraOptimizerConfig(mode="SE3")},
])
def test_unsupported_settings_are_rejected_loudly(kw):
    with pytest.raises(NotImplementedError):
        ThermalNerfactoModelConfig(**kw).validate_for_hip()
    ThermalNerfactoModelConfig().validate_for_hip()  # the reference defaults are inside the path


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_arena_layout_and_state_dict_contract(golden_dir, mode):
    cfg = ThermalNerfactoModelConfig(density_mode=mode)
    a = ParamArena(cfg, 8, "cpu")
    if mode == "shared":
        assert a.total == 22033792 and a.live_range == (0, 19411520)
    lo_prev = 0
    for g in ("proposal_networks", "fields", "camera_opt", "proposal_networks_thermal", "fields_thermal", "camera_opt_thermal"):
        lo, hi = a.group_range[g]
        assert lo == lo_prev and lo % 64 == 0 and hi >= lo
        lo_prev = hi
    assert lo_prev == a.total
    # every parameter of the reference's state_dict with a trainable counterpart is in the arena with the same shape, 256-byte aligned
    keys = json.load(open(os.path.join(golden_dir, f"state_dict_keys_{mode}.json")))
    tiny = {"hash_table"}  # golden files were written for tiny tables: compare every shape except the table row counts
    for name in a.names():
        off, shape = a.layout[name]
        assert off % 64 == 0
        assert name in keys, name
        ref_shape = keys[name][0]
        if any(t in name for t in tiny):
            assert list(shape)[1:] == ref_shape[1:]
        else:
            assert list(shape) == ref_shape, (name, shape, ref_shape)
    # views alias the arena
    name = a.names()[0]
    a.view(name).fill_(3.0)
    off, shape = a.layout[name]
    assert float(a.params[off]) == 3.0 and float(a.params[off + int(np.prod(shape)) - 1]) == 3.0
    a.grad_view(name).fill_(1.0)
    a.zero_grad()
    assert float(a.grads.abs().max()) == 0.0


def test_ray_containers():
    n = 10
    rb = RayBundle(origins=torch.arange(n * 3.0).view(n, 3), directions=torch.ones(n, 3), pixel_area=torch.ones(n, 1), camera_indices=torch.arange(n).view(n, 1))
    assert len(rb) == n and rb.shape == (n,)
    sl = rb[2:5]
    assert len(sl) == 3 and torch.equal(sl.origins, rb.origins[2:5])
    grid = RayBundle(origins=torch.arange(24.0).view(2, 4, 3), directions=torch.ones(2, 4, 3), pixel_area=torch.ones(2, 4, 1), camera_indices=torch.zeros(2, 4, 1))
    assert len(grid) == 8 and len(grid.flatten()) == 8
    part = grid.get_row_major_sliced_ray_bundle(3, 6)
    assert torch.equal(part.origins, grid.origins.reshape(-1, 3)[3:6])
    # the reference's one pinned value (tests/cameras/test_rays.py:11-30)
    fr = Frustums(origins=torch.tensor([[0.0, 1.0, 2.0]]), directions=torch.tensor([[0.0, 1.0, 0.0]]), starts=torch.tensor([[2.0]]), ends=torch.tensor([[3.0]]),
                  pixel_area=torch.ones(1, 1))
    assert fr.get_positions().reshape(-1).tolist() == pytest.approx([0.0, 3.5, 2.0], abs=1e-6)


def test_lr_schedule_and_optimizer_table():
    """ExponentialDecayScheduler (engine/schedulers.py:109-141) and the optimiser table of method_configs["thermal-nerfacto"]."""
    lr0, lr1, steps = OPTIMIZERS["fields"]
    assert (lr0, lr1, steps) == (1e-2, 1e-4, 200000)
    assert exp_decay_lr(0, lr0, lr1, steps) == pytest.approx(lr0) and exp_decay_lr(steps, lr0, lr1, steps) == pytest.approx(lr1)
    assert exp_decay_lr(steps // 2, lr0, lr1, steps) == pytest.approx((lr0 * lr1) ** 0.5)  # log-linear
    assert exp_decay_lr(10 * steps, lr0, lr1, steps) == pytest.approx(lr1)  # clamped
    assert set(OPTIMIZERS) >= {"proposal_networks", "fields", "camera_opt", "proposal_networks_thermal", "fields_thermal", "camera_opt_thermal"}
    assert OPTIMIZERS["camera_opt"][:2] == (1e-3, 1e-4)


def test_method_plugin_object():
    from nerfstudio_thermal_amd import plugin

    spec = plugin.thermal_nerfacto_hip
    cfg = spec.config.pipeline.model if hasattr(spec, "config") else spec.model
    assert isinstance(cfg, ThermalNerfactoModelConfig) and cfg.camera_optimizer.mode == "SO3xR3"
    cfg.validate_for_hip()


# (plugin._build() inside a nerfstudio installation is tested against the REAL reference classes: tests/test_real_trainer_cpu.py)


def test_uniform_pool_hands_out_fresh_disjoint_slices():
    """ops.UniformPool: one torch.rand per `steps` requests, slices never overlap, never repeat, stay valid across a refill, and the sequence
    is a function of the seed alone (what lets a test replay the data manager's draws)."""
    from nerfstudio_thermal_amd.ops import UniformPool

    torch.manual_seed(5)
    pool = UniformPool("cpu", steps=4)
    got = [pool.take((3, 50)) for _ in range(9)]  # 9 requests of 150 floats (192-float spans): refills after every 4
    for t in got:
        assert t.shape == (3, 50) and t.is_contiguous() and float(t.min()) >= 0.0 and float(t.max()) < 1.0
    ptrs = sorted((t.data_ptr(), t.data_ptr() + 4 * t.numel()) for t in got)
    assert all(a_end <= b for (_, a_end), (b, _) in zip(ptrs, ptrs[1:]))  # disjoint storage, also across refills (new allocations)
    keep = got[0].clone()
    pool.take((1000,))  # larger than a span: forces a refill of its own size
    assert torch.equal(got[0], keep)
    torch.manual_seed(5)
    again = UniformPool("cpu", steps=4)
    assert all(torch.equal(a, again.take((3, 50))) for a in got)


def test_fused_scaler_step_cuts_launches_at_optimiser_boundaries():
    """optim._cut_launches: the launches of Optimizers._fused_scaler_step (torch.amp.GradScaler.step for every optimiser at once).  A skipped
    step is counted once per flag and launch, so an optimiser may only be split when it alone exceeds a launch."""
    from nerfstudio_thermal_amd.optim import _cut_launches

    A, B = object(), object()
    h, h2 = (0.9, 0.999, 1e-15), (0.9, 0.99, 1e-15)

    def rng(flag, k, arena=A, hyper=h):
        return [(flag, arena, 100 * i, 100 * i + 64, 1, 1e-2, hyper) for i in range(k)]

    def flags(launches):
        return [[w[0] for w in ch] for ch in launches]

    # the normal case: one range per optimiser, one launch
    assert flags(_cut_launches(rng(0, 1) + rng(1, 1) + rng(2, 1))) == [[0, 1, 2]]
    # 3 + 3 + 3 ranges: the third optimiser does not fit behind the first two -> it starts the next launch whole
    assert flags(_cut_launches(rng(0, 3) + rng(1, 3) + rng(2, 3))) == [[0] * 3 + [1] * 3, [2] * 3]
    # an optimiser of 11 ranges is split (only its last launch counts), the next one rides with its tail
    assert flags(_cut_launches(rng(0, 11) + rng(1, 2))) == [[0] * 8, [0] * 3 + [1] * 2]
    # another arena / other betas start a launch of their own
    assert flags(_cut_launches(rng(0, 2) + rng(1, 2, arena=B) + rng(2, 1, arena=B, hyper=h2))) == [[0, 0], [1, 1], [2]]
    # every range goes out exactly once, in order
    work = rng(0, 5) + rng(1, 9) + rng(2, 1) + rng(3, 7)
    out = _cut_launches(work)
    assert [w for ch in out for w in ch] == work and all(len(ch) <= 8 for ch in out)
    for k, ch in enumerate(out[:-1]):  # a launch ends inside an optimiser only if that optimiser also started it
        if out[k + 1][0][0] == ch[-1][0]:
            assert ch[0][0] == ch[-1][0]
