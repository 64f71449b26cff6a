"""The DataManager adapter (nerfstudio_thermal_amd/datamanager.py) on the GPU: `next_train(step) -> (RayBundle, batch)` with the reference's
batch keys, every value consistent with the scene on disk (data/datamanagers/base_datamanager.py:538-547, data/pixel_samplers.py:296-337,389-441,
model_components/ray_generators.py:40-55)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.dataparser import write_rgbt_dataset

    cams = synth.synth_cameras()
    images = synth.synth_images(cams)
    d = str(tmp_path_factory.mktemp("dm") / "scene")
    write_rgbt_dataset(d, cams, images)
    return d


def test_next_train_is_the_references_pair(scene):
    from nerfstudio_thermal_amd import ops
    from nerfstudio_thermal_amd.datamanager import HipDataManager, HipDataManagerConfig
    from nerfstudio_thermal_amd.dataparser import ThermalNerfDataParserConfig
    from nerfstudio_thermal_amd.rays import RayBundle

    cfg = HipDataManagerConfig(data=scene, dataparser=ThermalNerfDataParserConfig(train_split_fraction=0.75), train_num_rays_per_batch=1026, eval_num_rays_per_batch=64)
    dm = cfg.setup(device="cuda:0", test_mode="val")
    assert isinstance(dm, HipDataManager) and dm.get_param_groups() == {} and dm.get_training_callbacks(None) == []
    n_img = len(dm.train_dataset)
    assert dm.get_train_rays_per_batch() == 1024  # PatchPixelSampler.set_num_rays_per_batch: a multiple of patch_size^2
    images = [dm.train_dataset.get_image_float32(i) for i in range(n_img)]
    is_th = dm.train_dataset.metadata["is_thermal"]
    cams = dm._train_src.cameras
    for step in range(3):
        rb, batch = dm.next_train(step)
        assert isinstance(rb, RayBundle) and set(batch) == {"image", "indices", "is_thermal"}
        idx = batch["indices"].cpu()
        assert idx.dtype == torch.int64 and tuple(idx.shape) == (1024, 3)
        assert tuple(rb.origins.shape) == (1024, 3) and tuple(rb.camera_indices.shape) == (1024, 1) and tuple(rb.pixel_area.shape) == (1024, 1)
        assert torch.equal(rb.camera_indices[:, 0].cpu(), idx[:, 0])
        # num_rays // num_images rays per image, image after image in batch order, the last image takes the remainder (pixel_samplers.py:296-312)
        per = (1024 // n_img // 4) * 4
        order = dm._train_src.batch_order
        expect_cam = sum(([c] * per for c in order[:-1]), []) + [order[-1]] * (1024 - per * (n_img - 1))
        assert idx[:, 0].tolist() == expect_cam
        # 2x2 patches: 4 consecutive rays = (y,x), (y,x+1), (y+1,x), (y+1,x+1)
        p = idx.view(-1, 4, 3)
        assert torch.equal(p[:, 1, 2], p[:, 0, 2] + 1) and torch.equal(p[:, 2, 1], p[:, 0, 1] + 1) and torch.equal(p[:, 3, 1:], p[:, 0, 1:] + 1)
        # ground truth and is_thermal are those of the indexed pixels / images
        gt = torch.stack([images[c][y, x] for c, y, x in idx.tolist()])
        assert torch.equal(batch["image"].cpu(), gt)
        assert batch["is_thermal"].cpu().tolist() == [float(is_th[c]) for c in idx[:, 0].tolist()]
        # the bundle is RayGenerator(indices) on the split's cameras
        o, d, area, nrm = ops.raygen(batch["indices"], cams["c2w"], cams["fx"], cams["fy"], cams["cx"], cams["cy"], cams["distortion"])
        assert torch.equal(rb.origins, o) and torch.equal(rb.directions, d) and torch.equal(rb.pixel_area, area)
        assert torch.equal(rb.metadata["directions_norm"], nrm)
    assert dm.train_count == 3
    # fresh pixels every step
    a, b = dm.next_train(3)[1]["indices"], dm.next_train(4)[1]["indices"]
    assert not torch.equal(a, b)
    # eval side: one image at a time, round robin; every image once through the fixed-indices view
    cam, eb = dm.next_eval_image(0)
    assert eb["image"].shape[:2] == (cam.height, cam.width) and eb["image"].device.type == "cuda"
    assert len(dm.fixed_indices_eval_dataloader) == len(dm.eval_dataset)
    assert dm.get_datapath().name == "scene"


def test_adapter_feeds_the_fused_trainer(scene):
    """trainer.fused_train_iteration reads pipeline.datamanager.next_train(step) -> the adapter's pair goes straight into the fused step."""
    import types

    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
    from nerfstudio_thermal_amd.datamanager import HipDataManagerConfig
    from nerfstudio_thermal_amd.dataparser import ThermalNerfDataParserConfig
    from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers
    from nerfstudio_thermal_amd.trainer import FusedTrainerMixin

    dm = HipDataManagerConfig(data=scene, dataparser=ThermalNerfDataParserConfig(train_split_fraction=0.9), train_num_rays_per_batch=256).setup(device="cuda:0")
    cfg = ThermalNerfactoModelConfig(density_mode="shared", log2_hashmap_size=12)
    for a in cfg.proposal_net_args_list:
        a["log2_hashmap_size"] = 10
    model = cfg.setup(scene_box=dm.train_dataset.scene_box, num_train_data=len(dm.train_dataset), metadata=dm.train_dataset.metadata, device="cuda:0")
    model.train()

    class _Base:
        def train_iteration(self, step):
            raise AssertionError("fell through to the reference iteration")

    class _T(FusedTrainerMixin, _Base):
        pass

    t = _T()
    t.pipeline = types.SimpleNamespace(model=model, datamanager=dm)
    t.optimizers = Optimizers(model.get_param_groups(), optimizer_cls=HipFusedAdam)
    t.mixed_precision, t.grad_scaler = True, torch.amp.GradScaler("cuda")
    t.gradient_accumulation_steps = {}
    t.config = types.SimpleNamespace(log_gradients=False)
    before = model.arena.params.clone()
    for step in range(3):
        loss, loss_dict, metrics = t.train_iteration(step)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and set(loss_dict) >= {"rgb_loss", "thermal_loss", "interlevel_loss", "distortion_loss"}
    assert dm.train_count == 3 and not torch.equal(before, model.arena.params)


def test_next_batch_is_sampled_inside_the_training_step(scene):
    """prefetch="cowork" (the default): the batch of iteration k + 1 is sampled in co-work blocks of iteration k's optimiser launch
    (TnTrainStep.next_sample) -- the same batches, bit for bit, as a manager that launches tn_sample_rays itself draws from the same seed, and
    the pending block is gone after the step (taken), not launched a second time."""
    import types

    from nerfstudio_thermal_amd import ops
    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
    from nerfstudio_thermal_amd.datamanager import HipDataManagerConfig
    from nerfstudio_thermal_amd.dataparser import ThermalNerfDataParserConfig
    from nerfstudio_thermal_amd.optim import HipFusedAdam, Optimizers
    from nerfstudio_thermal_amd.trainer import FusedTrainerMixin

    def manager(prefetch):
        import random

        random.seed(5)  # (the batch order is a random permutation of the images)
        dm = HipDataManagerConfig(data=scene, dataparser=ThermalNerfDataParserConfig(train_split_fraction=0.9), train_num_rays_per_batch=256).setup(device="cuda:0")
        dm._train_src._dm.prefetch = prefetch
        return dm

    dm = manager("cowork")
    cfg = ThermalNerfactoModelConfig(density_mode="shared", log2_hashmap_size=12)
    for a in cfg.proposal_net_args_list:
        a["log2_hashmap_size"] = 10
    model = cfg.setup(scene_box=dm.train_dataset.scene_box, num_train_data=len(dm.train_dataset), metadata=dm.train_dataset.metadata, device="cuda:0")
    model.train()
    seen = []

    class _Base:
        def train_iteration(self, step):
            raise AssertionError("fell through to the reference iteration")

    class _T(FusedTrainerMixin, _Base):
        pass

    t = _T()
    orig_next = dm.next_train

    def spy(step):
        rb, batch = orig_next(step)
        seen.append((rb.origins, rb.directions, rb.camera_indices, batch["image"], batch["indices"], batch["is_thermal"], rb.pixel_area))
        return rb, batch

    t.pipeline = types.SimpleNamespace(model=model, datamanager=types.SimpleNamespace(next_train=spy))
    t.optimizers = Optimizers(model.get_param_groups(), optimizer_cls=HipFusedAdam)
    t.mixed_precision, t.grad_scaler = True, torch.amp.GradScaler("cuda")
    t.gradient_accumulation_steps = {}
    t.config = types.SimpleNamespace(log_gradients=False)
    torch.manual_seed(23)  # (the manager's uniforms come from ONE torch.rand per 32 batches, drawn at its first batch)
    for step in range(4):
        t.train_iteration(step)
        assert ops._PENDING_SAMPLE is None, step  # the step's optimiser launch took the next batch
    torch.cuda.synchronize()
    ref = manager(False)
    torch.manual_seed(23)
    for step, got in enumerate(seen):
        rb, batch = ref.next_train(step)
        want = (rb.origins, rb.directions, rb.camera_indices, batch["image"], batch["indices"], batch["is_thermal"], rb.pixel_area)
        for a, b in zip(got, want):
            assert torch.equal(a, b), step



@pytest.mark.parametrize("mode,counts", [("cowork", None), ("serial", None), ("fork", None), ("fused", None),
                                         ("cowork", ((200, 80), 50)), ("fused", ((200, 80), 50))])  # (last trips of a level with 8 / 16 lanes; 50 x N field samples: partial tiles)
def test_next_sampling_inside_the_training_step(scene, monkeypatch, mode, counts):
    """TnTrainStep.next_sampling: the sampling front of iteration k + 1 (pose correction, level-0 bins, both proposal levels' density -> weights ->
    PDF resampling) runs in co-work blocks of iteration k's optimiser launch, behind a launch that steps the groups it reads.  Against the same run
    with every iteration sampling in line (engine.next_sampling = False), from re-synchronised states as the other comparisons here: the forward's
    buffer -- pose-corrected rays, all bins, proposal densities / weights / median depths, and everything the field and the renderers add -- bit for
    bit, over 13 iterations that include proposal updates, iterations without one and a skipped (forced-inf) one; the same losses, scaler and
    sampler bookkeeping.  mode "serial" (TN_NEXT_SAMPLING=2): the chain as a launch of its own behind the optimiser launch (the A/B timing aid)."""
    import random

    from nerfstudio_thermal_amd import ops
    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
    from nerfstudio_thermal_amd.datamanager import HipDataManagerConfig
    from nerfstudio_thermal_amd.dataparser import ThermalNerfDataParserConfig
    from nerfstudio_thermal_amd.optim import DeviceGradScaler

    N = 256
    if mode != "cowork":  # (A/B timing aids: the chain as a launch of its own behind the optimiser launch / on a companion stream beside it)
        # ("fused", TN_NEXT_SAMPLING=4: the chain's waves also step most of the field's optimiser range between their stages)
        monkeypatch.setenv("TN_NEXT_SAMPLING", {"serial": "2", "fork": "3", "fused": "4"}[mode])

    def run(chain: bool, states=None):
        random.seed(5)
        torch.manual_seed(29)
        dm = HipDataManagerConfig(data=scene, dataparser=ThermalNerfDataParserConfig(train_split_fraction=0.9), train_num_rays_per_batch=N).setup(device="cuda:0")
        extra = {} if counts is None else {"num_proposal_samples_per_ray": counts[0], "num_nerf_samples_per_ray": counts[1]}
        cfg = ThermalNerfactoModelConfig(density_mode="shared", log2_hashmap_size=12, **extra)
        for a in cfg.proposal_net_args_list:
            a["log2_hashmap_size"] = 10
        model = cfg.setup(scene_box=dm.train_dataset.scene_box, num_train_data=len(dm.train_dataset), metadata=dm.train_dataset.metadata, device="cuda:0")
        model.train()
        eng, ar = model.engine, model.arena
        eng.next_sampling = chain
        scaler = DeviceGradScaler("cuda:0", num_groups=len(ar.optimised_groups))
        rec, used = [], 0
        for step in range(13):
            rb, batch = dm.next_train(step)
            if step == 6:
                batch["image"].fill_(float("inf"))  # in place: the batch tensors are the ones the previous iteration's co-work filled
            if states is not None:  # the in-line run continues from the chained run's state (float atomics: two runs drift apart in the last bits)
                with torch.no_grad():
                    for dst, src in zip((ar.params, ar.exp_avg, ar.exp_avg_sq), states[step]):
                        dst.copy_(src)
            before = (ar.params.clone(), ar.exp_avg.clone(), ar.exp_avg_sq.clone())
            planned = eng.__dict__.get("_planned") is not None
            losses = model.train_iteration(rb, batch, step, grad_scaler=scaler)
            call = eng._step_call
            used += int(planned and call.st.sampling_done == 1)
            torch.cuda.synchronize()
            rec.append({"before": before, "after": (ar.params.clone(), ar.exp_avg.clone(), ar.exp_avg_sq.clone()), "buf": call._keep[3].clone(), "off": list(call.off), "losses": {k: float(v) for k, v in losses.items()},
                        "updated": eng.last_updated, "since": eng.steps_since_update, "scale": scaler.get_scale(),
                        "skipped": [scaler.num_skipped(i) for i in range(3)], "lag": scaler.schedule_lag(), "counts": call.counts})
            assert ops._PENDING_SAMPLE is None, step
        return rec, used

    A, usedA = run(True)
    assert usedA == 12  # every iteration but the first started at the field
    B, usedB = run(False, [r["before"] for r in A])
    assert usedB == 0
    saw_plain = saw_update = False
    for step, (a, b) in enumerate(zip(A, B)):
        assert a["off"] == b["off"]
        S0, S1, S2 = a["counts"]
        sizes = [N * 3, N * 3, N * (S0 + 1), N * (S0 + 1), N * S0, N * S0, N, N * (S1 + 1), N * (S1 + 1), N * S1, N * S1, N, N * (S2 + 1), N * (S2 + 1), N * S2,
                 N * S2, N * S2 * 4, N * 4, N, N, N]
        for slot, size in enumerate(sizes):
            o = a["off"][slot]
            assert torch.equal(a["buf"][o:o + size], b["buf"][o:o + size]), (step, slot)
        if a["updated"]:  # the proposal levels' encodings are kept on update iterations
            for slot, size in ((22, N * S0 * 10), (23, N * S1 * 10)):
                o = a["off"][slot]
                assert torch.equal(a["buf"][o:o + size], b["buf"][o:o + size]), (step, slot)
        saw_plain |= not a["updated"]
        saw_update |= a["updated"] and step > 0
        for k in ("updated", "since", "scale", "skipped", "lag"):
            assert a[k] == b[k], (step, k)
        if step != 6:
            for k, v in a["losses"].items():
                assert abs(v - b["losses"][k]) <= 1e-5 * abs(b["losses"][k]) + 1e-12, (step, k)
        # the optimiser step itself, whichever launch layout carried it: from the same state and (up to the order of float atomics) the same
        # gradients -- parameters within a thousandth of the largest learning rate (1e-2: a batch stepped twice or not at all is off by a whole
        # step), moments to 1e-4, and the entries that never saw a gradient untouched in both
        for which, (x, y) in enumerate(zip(a["after"], b["after"])):
            tol = 1e-5 if which == 0 else 1e-4 * float(y.abs().max()) + 1e-12
            assert float((x - y).abs().max()) <= tol, (step, which, float((x - y).abs().max()), tol)
            if which > 0:
                assert float(((x == 0) != (y == 0)).float().mean()) <= 1e-4, (step, which)  # (an exact cancellation may differ in one of the runs)
    assert saw_plain and saw_update and A[-1]["skipped"][1] == 1
