"""Method-plugin entry (SURVEY.md 8b tier 1) for nerfstudio's method registry (plugins/registry.py:34-79, plugins/types.py:23-33).  Use with
    NERFSTUDIO_METHOD_CONFIGS="thermal-nerfacto-hip=nerfstudio_thermal_amd.plugin:thermal_nerfacto_hip"
and `ns-train thermal-nerfacto-hip --data ...` takes the flags of `ns-train thermal-nerfacto` unchanged (README.md:108-128 of the reference).

The method NAME.  north_star's drop-in is `ns-train thermal-nerfacto`.  The reference's registry merges DISCOVERED methods over its in-tree table
(configs/method_configs.py:785-787: merge_methods(..., *discover_methods()) with the default overwrite=True), keyed by the name given in the
environment variable (plugins/registry.py:59-73) or by `config.method_name` for an entry point (:50).  So
    NERFSTUDIO_METHOD_CONFIGS="thermal-nerfacto=nerfstudio_thermal_amd.plugin:thermal_nerfacto_hip"  TN_METHOD_NAME=thermal-nerfacto
makes `ns-train thermal-nerfacto` itself run on this package (TN_METHOD_NAME sets `config.method_name`, i.e. the output directory and the
entry-point key); by default the specification is named `thermal-nerfacto-hip` so that both can be compared side by side.

When nerfstudio itself is importable the full TrainerConfig of method_configs["thermal-nerfacto"] (configs/method_configs.py:255-310) is
rebuilt with this package's model config, datamanager (datamanager.py: pixel sampling + ray generation on the device) and Trainer class
(trainer.FusedTrainerMixin ahead of the reference Trainer); otherwise a plain description object carrying the same optimiser table is exposed.

`HipTrainer` and `HipVanillaDataManager` are MODULE attributes: the reference writes config.yml with yaml.dump (configs/experiment_config.py:137:
`!!python/name:nerfstudio_thermal_amd.plugin.HipTrainer`) and ns-eval / ns-viewer / ns-render / ns-export read it back with yaml.load
(utils/eval_utils.py:89), which resolves the name by importing this module (tests/test_real_trainer_cpu.py round-trips the config)."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict

from .config import CameraOptimizerConfig, ThermalNerfactoModelConfig
from .engine import OPTIMIZERS

DEFAULT_METHOD_NAME = "thermal-nerfacto-hip"


def method_name() -> str:
    return os.environ.get("TN_METHOD_NAME", DEFAULT_METHOD_NAME)


def model_config() -> ThermalNerfactoModelConfig:
    return ThermalNerfactoModelConfig(eval_num_rays_per_chunk=1 << 15, camera_optimizer=CameraOptimizerConfig(mode="SO3xR3"))


@dataclass
class MethodDescription:
    method_name: str = field(default_factory=method_name)
    description: str = "thermal-nerfacto (RGB+thermal NeRF) on the MI355X-native HIP hot path"
    max_num_iterations: int = 30000
    steps_per_eval_batch: int = 500
    steps_per_save: int = 2000
    train_num_rays_per_batch: int = 8192
    eval_num_rays_per_batch: int = 8192
    patch_size: int = 2
    model: ThermalNerfactoModelConfig = field(default_factory=model_config)
    optimizers: Dict[str, Any] = field(default_factory=lambda: {k: {"lr": v[0], "eps": 1e-15, "lr_final": v[1], "max_steps": v[2]} for k, v in OPTIMIZERS.items()})


try:  # only the ABSENCE of nerfstudio selects the stand-alone description; any other failure is the user's to see
    from nerfstudio.engine.trainer import Trainer as _ReferenceTrainer
except ImportError:
    _ReferenceTrainer = None

if _ReferenceTrainer is not None:
    from .datamanager import make_nerfstudio_datamanager
    from .trainer import FusedTrainerMixin

    class HipTrainer(FusedTrainerMixin, _ReferenceTrainer):
        """TrainerConfig._target (engine/trainer.py:56): the reference Trainer with its train_iteration on the fused step (trainer.py); loop,
        callbacks, logging, evaluation, checkpoints and viewer stay the reference's."""

    HipVanillaDataManager = make_nerfstudio_datamanager()
    HipVanillaDataManager.__module__ = __name__  # addressable as nerfstudio_thermal_amd.plugin.HipVanillaDataManager (yaml, pickle)


def _build():
    """Inside a nerfstudio installation: a real MethodSpecification (plugins/types.py:23-33) whose TrainerConfig is a copy of
    method_configs["thermal-nerfacto"] (configs/method_configs.py:255-310) with this package's model config, datamanager class and Trainer class
    swapped in.  A method table without "thermal-nerfacto", or a config of another shape, is an error the user must see, not a silent fallback.
    Switches: TN_FUSED_TRAINER=0 keeps the unmodified Trainer, TN_DEVICE_DATAMANAGER=0 the reference's VanillaDataManager (host-side sampling)."""
    if _ReferenceTrainer is None:
        return MethodDescription()
    import copy

    from nerfstudio.configs.method_configs import method_configs
    from nerfstudio.plugins.types import MethodSpecification

    base = copy.deepcopy(method_configs["thermal-nerfacto"])
    base.method_name = method_name()
    base.pipeline.model = model_config()
    if os.environ.get("TN_DEVICE_DATAMANAGER", "1") != "0":
        base.pipeline.datamanager._target = HipVanillaDataManager
    if os.environ.get("TN_FUSED_TRAINER", "1") != "0":
        base._target = HipTrainer
    return MethodSpecification(config=base, description=MethodDescription().description)


thermal_nerfacto_hip = _build()

if os.environ.get("TN_SINGLE_THREAD_BACKWARD", "0") == "1":  # opt-in (see configure_host): the training process that loads this plugin
    from . import configure_host as _configure_host

    _configure_host(single_thread_backward=True)
