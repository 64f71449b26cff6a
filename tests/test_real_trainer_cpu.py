"""The method plugin, trainer.FusedTrainerMixin and the device datamanager against the REFERENCE's own classes (build container only: the
reference is imported through oracle/ref_import.py with permissive stubs for the third-party packages this image lacks; skipped where
/root/reference is absent, i.e. on the GPU box).  No GPU: nothing here launches a kernel.

  * the mixin lands AHEAD of the real `nerfstudio.engine.trainer.Trainer` in the MRO and overrides exactly train_iteration / save_checkpoint /
    _load_checkpoint with call-compatible signatures (engine/trainer.py:389,425,456);
  * every attribute the mixin reads exists on a constructed Trainer or is assigned by Trainer.setup (engine/trainer.py:112-140,142-156);
  * plugin._build() yields a MethodSpecification (plugins/types.py:23-33) whose config._target is the mixed class and whose datamanager target is
    a VanillaDataManager subclass, with the reference's optimiser table and flags untouched;
  * the config survives the reference's own persistence: yaml.dump (configs/experiment_config.py:137) -> yaml.load(Loader=yaml.Loader)
    (utils/eval_utils.py:89) resolves both classes by name;
  * the datamanager subclass is built by the reference's config machinery on a scene parsed by the reference's ThermalNerf dataparser, keeps the
    reference's datasets / eval loaders, and REFUSES a CPU device loudly (no host fallback for the sampler).
"""
import inspect
import os
import re
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")


@pytest.fixture(scope="module")
def ns():
    trainer_cls, trainer_cfg, method_configs, method_spec = ref_import.import_reference_trainer()
    sys.modules.pop("nerfstudio_thermal_amd.plugin", None)  # (re)import with nerfstudio importable
    import nerfstudio_thermal_amd.plugin as plugin

    return dict(Trainer=trainer_cls, TrainerConfig=trainer_cfg, method_configs=method_configs, MethodSpecification=method_spec, plugin=plugin)


def test_mixin_resolves_ahead_of_the_real_trainer(ns):
    from nerfstudio_thermal_amd.trainer import FusedTrainerMixin

    Trainer, Hip = ns["Trainer"], ns["plugin"].HipTrainer
    assert Hip.__mro__[:3] == (Hip, FusedTrainerMixin, Trainer)
    for name in ("train_iteration", "save_checkpoint", "_load_checkpoint"):
        assert getattr(Hip, name) is getattr(FusedTrainerMixin, name), name
        assert name in vars(Trainer), f"the reference Trainer no longer defines {name}"
        ours = list(inspect.signature(getattr(FusedTrainerMixin, name)).parameters)
        # (the reference decorates these with @profiler.time_function / @check_main_thread, which hide the signature: read the def line)
        m = re.search(rf"def {name}\(([^)]*)\)", inspect.getsource(inspect.getmodule(Trainer)))
        ref = [a.split(":")[0].strip() for a in m.group(1).split(",") if a.strip()]
        assert ours == ref, (name, ours, ref)
    # everything else stays the reference's
    for name in ("train", "setup", "setup_optimizers", "eval_iteration", "_init_viewer_state", "_update_viewer_state"):
        assert getattr(Hip, name) is getattr(Trainer, name), name
    # same class when mixed by hand, as the module docstring shows
    assert type("HipTrainer", (FusedTrainerMixin, Trainer), {}).train_iteration is FusedTrainerMixin.train_iteration


def test_attributes_the_mixin_reads_exist_on_a_constructed_trainer(ns, tmp_path):
    import copy

    cfg = copy.deepcopy(ns["plugin"].thermal_nerfacto_hip.config)
    cfg.machine.device_type = "cpu"
    cfg.output_dir = tmp_path
    cfg.set_timestamp()
    t = cfg.setup(local_rank=0, world_size=1)  # InstantiateConfig.setup -> _target(config, ...)
    assert type(t) is ns["plugin"].HipTrainer
    assert hasattr(t, "mixed_precision") and hasattr(t, "grad_scaler") and callable(t.grad_scaler.is_enabled)
    assert t.gradient_accumulation_steps["fields"] == 1 and list(t.gradient_accumulation_steps.values()) in ([], [1])
    assert hasattr(t.config, "log_gradients") and t.config.log_gradients is False
    src = inspect.getsource(ns["Trainer"].setup)
    assert "self.pipeline = " in src and "self.optimizers = " in src  # assigned by setup(), which needs the GPU model
    # fused_ready() on a Trainer whose pipeline holds a foreign model falls through to the reference's iteration
    import types

    from nerfstudio_thermal_amd.trainer import fused_ready

    t.pipeline = types.SimpleNamespace(model=object())
    assert not fused_ready(t)
    ours = types.SimpleNamespace(engine=object(), train_iteration=lambda *a, **k: None)
    t.pipeline = types.SimpleNamespace(model=ours)
    assert fused_ready(t)


def test_plugin_builds_a_method_specification(ns):
    plugin, ref = ns["plugin"], ns["method_configs"]["thermal-nerfacto"]
    spec = plugin.thermal_nerfacto_hip
    assert isinstance(spec, ns["MethodSpecification"]) and isinstance(spec.config, ns["TrainerConfig"])
    c = spec.config
    assert c._target is plugin.HipTrainer and c.method_name == "thermal-nerfacto-hip"
    from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManager

    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig

    assert c.pipeline.datamanager._target is plugin.HipVanillaDataManager and issubclass(plugin.HipVanillaDataManager, VanillaDataManager)
    assert isinstance(c.pipeline.model, ThermalNerfactoModelConfig)
    # the reference's table is untouched and everything else is the reference's entry
    assert ref._target is ns["Trainer"] and ref.method_name == "thermal-nerfacto"
    assert set(c.optimizers) == set(ref.optimizers)
    for k in ("max_num_iterations", "steps_per_save", "steps_per_eval_batch", "mixed_precision"):
        assert getattr(c, k) == getattr(ref, k), k
    dm, rdm = c.pipeline.datamanager, ref.pipeline.datamanager
    assert (dm.train_num_rays_per_batch, dm.eval_num_rays_per_batch, dm.pixel_sampler.patch_size) == (rdm.train_num_rays_per_batch, rdm.eval_num_rays_per_batch, 2)
    # every flag README.md:108-128 documents exists on the model config with the reference's default
    for flag in ("density_mode", "density_loss_mult", "rgb_density_loss_mult", "cross_channel_loss_mult", "thermal_loss_mult", "tv_pixel_loss_mult",
                 "removal_min_density_diff"):
        assert getattr(c.pipeline.model, flag) == getattr(ref.pipeline.model, flag), flag


def test_a_method_table_without_the_method_is_an_error_not_a_fallback(ns, monkeypatch):
    import nerfstudio.configs.method_configs as mc

    table = dict(ns["method_configs"])
    del table["thermal-nerfacto"]
    monkeypatch.setattr(mc, "method_configs", table)
    with pytest.raises(KeyError):
        ns["plugin"]._build()
    # and _build() copies: the reference's entry is never mutated
    monkeypatch.setattr(mc, "method_configs", ns["method_configs"])
    ref = ns["method_configs"]["thermal-nerfacto"]
    spec = ns["plugin"]._build()
    assert spec.config is not ref and spec.config.pipeline is not ref.pipeline and type(ref.pipeline.model).__module__.startswith("nerfstudio.models")


def test_switches_and_the_in_tree_name(ns, monkeypatch):
    plugin = ns["plugin"]
    monkeypatch.setenv("TN_FUSED_TRAINER", "0")
    monkeypatch.setenv("TN_DEVICE_DATAMANAGER", "0")
    monkeypatch.setenv("TN_METHOD_NAME", "thermal-nerfacto")
    spec = plugin._build()
    ref = ns["method_configs"]["thermal-nerfacto"]
    assert spec.config._target is ns["Trainer"] and spec.config.pipeline.datamanager._target == ref.pipeline.datamanager._target
    assert spec.config.method_name == "thermal-nerfacto"
    # the registry lets a discovered method take the in-tree name (configs/method_configs.py:785-787: overwrite defaults to True)
    from nerfstudio.configs.method_configs import merge_methods

    merged, _ = merge_methods({"thermal-nerfacto": ref}, {"thermal-nerfacto": ""}, {"thermal-nerfacto": spec.config}, {"thermal-nerfacto": spec.description})
    assert merged["thermal-nerfacto"] is spec.config
    # ... and the reference's own discovery finds the specification through the environment variable (plugins/registry.py:53-74)
    from nerfstudio.plugins.registry import discover_methods

    monkeypatch.setenv("NERFSTUDIO_METHOD_CONFIGS", "thermal-nerfacto=nerfstudio_thermal_amd.plugin:thermal_nerfacto_hip")
    methods, descriptions = discover_methods()
    assert methods["thermal-nerfacto"] is plugin.thermal_nerfacto_hip.config and "MI355X" in descriptions["thermal-nerfacto"]


def test_config_round_trips_through_the_references_yaml(ns):
    import yaml

    cfg = ns["plugin"].thermal_nerfacto_hip.config
    text = yaml.dump(cfg)
    assert "nerfstudio_thermal_amd.plugin.HipTrainer" in text and "nerfstudio_thermal_amd.plugin.HipVanillaDataManager" in text
    back = yaml.load(text, Loader=yaml.Loader)
    assert back._target is ns["plugin"].HipTrainer
    assert back.pipeline.datamanager._target is ns["plugin"].HipVanillaDataManager
    assert type(back.pipeline.model) is type(cfg.pipeline.model) and back.pipeline.model.density_mode == cfg.pipeline.model.density_mode
    assert back.method_name == cfg.method_name and set(back.optimizers) == set(cfg.optimizers)


def test_datamanager_subclass_on_a_scene_parsed_by_the_reference(ns, tmp_path):
    import copy

    import numpy as np
    import torch

    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.datamanager import camera_tensors
    from nerfstudio_thermal_amd.dataparser import ThermalNerfDataParserConfig, write_rgbt_dataset

    cams = synth.synth_cameras()
    images = synth.synth_images(cams)
    scene = str(tmp_path / "scene")
    write_rgbt_dataset(scene, cams, images)
    plugin = ns["plugin"]
    dm_cfg = copy.deepcopy(plugin.thermal_nerfacto_hip.config.pipeline.datamanager)
    dm_cfg.data = tmp_path / "scene"
    dm_cfg.train_num_rays_per_batch = 256
    # inference mode: datasets are built, setup_train / setup_eval are not run (base_datamanager.py:176-181) -> no GPU needed
    dm = dm_cfg.setup(device="cpu", test_mode="inference", world_size=1, local_rank=0)
    assert type(dm) is plugin.HipVanillaDataManager
    from nerfstudio.data.datasets.thermal_dataset import ThermalDataset

    assert type(dm.train_dataset) is ThermalDataset and len(dm.train_dataset) + len(dm.eval_dataset) == len(images)
    assert dm.get_param_groups() == {} and dm.get_datapath() == tmp_path / "scene"
    # the camera tensors the device sampler would use == this package's dataparser on the same scene (which tests/test_dataparser_cpu.py pins)
    ours = ThermalNerfDataParserConfig(data=scene).setup().get_dataparser_outputs("train")
    ct, co = camera_tensors(dm.train_dataset.cameras, "cpu"), camera_tensors(ours.cameras, "cpu")
    for k in ct:
        assert ct[k].dtype == torch.float32 and ct[k].is_contiguous()
        assert float((ct[k] - co[k]).abs().max()) <= 2e-6, k
    assert [float(x) for x in dm.train_dataset.metadata["is_thermal"]] == [float(x) for x in ours.metadata["is_thermal"]]
    im = dm.train_dataset.get_image_float32(0)
    assert im.dtype == torch.float32 and im.shape[2] == 3 and np.isfinite(im.numpy()).all()
    # the train side is the device path: on a CPU device it must fail, not fall back to the host sampler
    with pytest.raises(RuntimeError, match="needs a GPU"):
        dm.setup_train()
    dm_cfg2 = copy.deepcopy(dm_cfg)
    dm_cfg2.train_num_images_to_sample_from = 4
    dm2 = dm_cfg2.setup(device="cpu", test_mode="inference")
    with pytest.raises(NotImplementedError, match="ALL training images"):
        dm2.setup_train()
