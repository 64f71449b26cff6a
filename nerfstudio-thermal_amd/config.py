"""Model configuration: a source-compatible mirror of the reference's ThermalNerfactoModelConfig
(models/thermal_nerfacto.py:32-64 over models/nerfacto.py:52-133 over models/base_model.py ModelConfig).
Field names, defaults and meanings are the reference's; `_target` resolves to this package's model."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Literal, Optional, Tuple, Type


@dataclass
class CameraOptimizerConfig:
    """cameras/camera_optimizers.py:39-56."""

    mode: Literal["off", "SO3xR3", "SE3", "shared_SO3xR3"] = "off"
    trans_l2_penalty: float = 1e-2
    rot_l2_penalty: float = 1e-3
    penalty_scale: float = 1
    optimizer: Optional[object] = None
    scheduler: Optional[object] = None


def _default_prop_args() -> List[Dict]:
    return [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False},
    ]


@dataclass
class ThermalNerfactoModelConfig:
    _target: Optional[Type] = None  # filled in by model.py
    # ModelConfig (models/base_model.py)
    enable_collider: bool = True
    collider_params: Optional[Dict[str, float]] = field(default_factory=lambda: {"near_plane": 2.0, "far_plane": 6.0})
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
    eval_num_rays_per_chunk: int = 4096  # the class default; method_configs["thermal-nerfacto"] (and plugin.py) set 1 << 15
    prompt: Optional[str] = None
    # NerfactoModelConfig
    near_plane: float = 0.05
    far_plane: float = 1000.0
    background_color: Literal["random", "last_sample", "black", "white"] = "last_sample"
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    hidden_dim_transient: int = 64
    num_levels: int = 16
    base_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = field(default_factory=_default_prop_args)
    proposal_initial_sampler: Literal["piecewise", "uniform"] = "piecewise"
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    orientation_loss_mult: float = 0.0001
    pred_normal_loss_mult: float = 0.001
    use_proposal_weight_anneal: bool = True
    use_appearance_embedding: bool = True
    use_average_appearance_embedding: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    predict_normals: bool = False
    disable_scene_contraction: bool = False
    use_gradient_scaling: bool = False
    implementation: Literal["tcnn", "torch", "hip"] = "hip"
    appearance_embed_dim: int = 32
    average_init_density: float = 1.0
    camera_optimizer: CameraOptimizerConfig = field(default_factory=lambda: CameraOptimizerConfig(mode="SO3xR3"))
    # ThermalNerfactoModelConfig
    density_loss_mult: float = 5e-5
    density_mode: Literal["rgb_only", "shared", "separate"] = "separate"
    rgb_density_loss_mult: float = 0.01
    thermal_loss_mult: float = 100.0
    tv_rgb_loss_mult: float = 0
    tv_thermal_loss_mult: float = 0
    num_density_tv_samples: int = 5000
    tv_pixel_loss_mult: float = 1e-6
    cross_channel_loss_mult: float = 1e-6
    removal_min_density_diff: float = 0.05
    use_proposal_thermal_weight_anneal: bool = False
    camera_optimizer_thermal: CameraOptimizerConfig = field(default_factory=lambda: CameraOptimizerConfig(mode="SO3xR3", penalty_scale=10))
    shared_camera_optimizer: CameraOptimizerConfig = field(default_factory=lambda: CameraOptimizerConfig(mode="shared_SO3xR3", penalty_scale=-1))
    shared_camera_optimizer_thermal: CameraOptimizerConfig = field(default_factory=lambda: CameraOptimizerConfig(mode="shared_SO3xR3", penalty_scale=-1))

    def setup(self, **kwargs):
        """InstantiateConfig.setup (configs/base_config.py:47-54)."""
        target = self._target
        if target is None:
            from .model import ThermalNerfactoModel as target  # noqa: N813
        return target(self, **kwargs)

    # ------------------------------------------------------------------ what the HIP path supports
    def validate_for_hip(self) -> None:
        bad = []
        if self.density_mode not in ("shared", "separate"):
            bad.append("density_mode must be 'shared' or 'separate' (rgb_only hard-codes .to('cuda') in the reference and is outside the scoped configs)")
        if self.predict_normals:
            bad.append("predict_normals")
        if self.disable_scene_contraction:
            bad.append("disable_scene_contraction")
        if self.use_gradient_scaling:
            bad.append("use_gradient_scaling")
        if self.background_color != "last_sample":
            bad.append("background_color != 'last_sample'")
        if self.use_same_proposal_network:
            bad.append("use_same_proposal_network")
        if self.proposal_initial_sampler != "piecewise":
            bad.append("proposal_initial_sampler != 'piecewise'")
        if not self.use_single_jitter:
            bad.append("use_single_jitter=False")
        if self.tv_rgb_loss_mult > 0 or self.tv_thermal_loss_mult > 0:
            bad.append("density TV losses (reference implementation is .cuda()-only and off by default)")
        if self.num_levels != 16 or self.features_per_level != 2 or self.hidden_dim != 64 or self.hidden_dim_color != 64:
            bad.append("main field must be 16 levels x 2 features, hidden 64/64")
        if self.appearance_embed_dim != 32:
            bad.append("appearance_embed_dim != 32")
        if self.num_proposal_iterations != 2 or len(self.num_proposal_samples_per_ray) != 2:
            bad.append("exactly two proposal iterations")
        for a in self.proposal_net_args_list:
            if a.get("num_levels", 8) != 5 or a.get("hidden_dim", 64) != 16 or a.get("use_linear", False):
                bad.append("proposal nets must be 5 levels, hidden 16, use_linear=False")
        if max(self.num_proposal_samples_per_ray + (self.num_nerf_samples_per_ray,)) > 256:
            bad.append("at most 256 samples per ray per level")
        for name in ("shared_camera_optimizer", "shared_camera_optimizer_thermal"):
            co = getattr(self, name)
            if co.mode != "off" and co.penalty_scale >= 0:
                bad.append(f"{name} enabled (reference default is off)")
        for name in ("camera_optimizer", "camera_optimizer_thermal"):
            if getattr(self, name).mode not in ("off", "SO3xR3"):
                bad.append(f"{name}.mode must be 'off' or 'SO3xR3'")
        if bad:
            raise NotImplementedError("configuration outside the HIP hot path: " + "; ".join(bad))
