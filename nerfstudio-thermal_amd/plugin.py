"""Method-plugin entry (SURVEY.md 8b tier 1): `thermal-nerfacto-hip` for nerfstudio's method registry
(plugins/registry.py:34-79, plugins/types.py:23-33).  Use with
    NERFSTUDIO_METHOD_CONFIGS="thermal-nerfacto-hip=nerfstudio_thermal_amd.plugin:thermal_nerfacto_hip"
When nerfstudio itself is importable the full TrainerConfig of method_configs["thermal-nerfacto"] (configs/method_configs.py:255-310) is
rebuilt with this package's model config; otherwise a plain description object carrying the same optimiser table is exposed."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict

from .config import CameraOptimizerConfig, ThermalNerfactoModelConfig
from .engine import OPTIMIZERS


def model_config() -> ThermalNerfactoModelConfig:
    return ThermalNerfactoModelConfig(eval_num_rays_per_chunk=1 << 15, camera_optimizer=CameraOptimizerConfig(mode="SO3xR3"))


@dataclass
class MethodDescription:
    method_name: str = "thermal-nerfacto-hip"
    description: str = "thermal-nerfacto (RGB+thermal NeRF) on the MI355X-native HIP hot path"
    max_num_iterations: int = 30000
    steps_per_eval_batch: int = 500
    steps_per_save: int = 2000
    train_num_rays_per_batch: int = 8192
    eval_num_rays_per_batch: int = 8192
    patch_size: int = 2
    model: ThermalNerfactoModelConfig = field(default_factory=model_config)
    optimizers: Dict[str, Any] = field(default_factory=lambda: {k: {"lr": v[0], "eps": 1e-15, "lr_final": v[1], "max_steps": v[2]} for k, v in OPTIMIZERS.items()})


def _build():
    """Inside a nerfstudio installation: a real MethodSpecification (plugins/types.py:23-33) whose TrainerConfig is a copy of
    method_configs["thermal-nerfacto"] (configs/method_configs.py:255-310) with this package's model config swapped in.  Only the ABSENCE of
    nerfstudio (ImportError) selects the stand-alone description; anything else -- a method table without "thermal-nerfacto", a config of another
    shape -- is an error the user must see, not a silent fallback."""
    try:
        from nerfstudio.configs.method_configs import method_configs
        from nerfstudio.plugins.types import MethodSpecification
    except ImportError:
        return MethodDescription()
    import copy

    base = copy.deepcopy(method_configs["thermal-nerfacto"])
    base.method_name = "thermal-nerfacto-hip"
    base.pipeline.model = model_config()
    if os.environ.get("TN_FUSED_TRAINER", "1") != "0":
        # TrainerConfig._target (engine/trainer.py:56): the reference Trainer with its train_iteration on the fused step (trainer.py); loop,
        # callbacks, logging, evaluation, checkpoints and viewer stay the reference's.  TN_FUSED_TRAINER=0: the unmodified Trainer.
        from nerfstudio.engine.trainer import Trainer

        from .trainer import FusedTrainerMixin

        base._target = type("HipTrainer", (FusedTrainerMixin, Trainer), {"__doc__": FusedTrainerMixin.__doc__})
    return MethodSpecification(config=base, description=MethodDescription().description)


thermal_nerfacto_hip = _build()

import os as _os  # noqa: E402

if _os.environ.get("TN_SINGLE_THREAD_BACKWARD", "0") == "1":  # opt-in (see configure_host): the training process that loads this plugin
    from . import configure_host as _configure_host

    _configure_host(single_thread_backward=True)
