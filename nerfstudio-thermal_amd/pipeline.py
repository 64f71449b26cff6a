"""Train / eval driver over an RGB+T dataset on disk: what VanillaPipeline does for `ns-train thermal-nerfacto` and `ns-eval`
(pipelines/base_pipeline.py:230-470) with this package's pieces.

  dataset  -> dataparser.ThermalNerf (train / val splits) -> images resident in HBM (ops.ImageCache) + camera tensors
  train    -> data.DeviceDataManager.next_train (pixel sampling + ray generation on the device) -> ThermalNerfactoModel.train_iteration
  eval     -> for every image of the val split: Model.get_outputs_for_camera (chunked) -> get_image_metrics_and_images;
              get_average_eval_image_metrics returns the mean of every metric over the images that report it (PSNR / SSIM per spectrum)
"""
from __future__ import annotations

import types
from typing import Dict, List, Optional

import torch

from . import ops
from .config import ThermalNerfactoModelConfig
from .data import DeviceDataManager
from .dataparser import DataparserOutputs, ThermalNerfDataParserConfig, load_image_float32
from .rays import RayBundle


class ThermalPipeline:
    def __init__(self, data: str, model_config: Optional[ThermalNerfactoModelConfig] = None, device="cuda", num_rays_per_batch: int = 4096,
                 patch_size: int = 2, parser_config: Optional[ThermalNerfDataParserConfig] = None, seed: int = 0, mixed_precision: bool = True):
        from .model import SceneBox

        self.device = torch.device(device)
        pc = parser_config or ThermalNerfDataParserConfig(data=data)
        pc.data = data
        self.train_outputs: DataparserOutputs = pc.setup().get_dataparser_outputs("train")
        self.eval_outputs: DataparserOutputs = pc.setup().get_dataparser_outputs("val")
        tr = self.train_outputs
        cfg = model_config or ThermalNerfactoModelConfig(density_mode="shared")
        self.model = cfg.setup(scene_box=SceneBox(aabb=tr.scene_box_aabb), num_train_data=len(tr.image_filenames),
                               metadata={"is_thermal": list(tr.metadata["is_thermal"])}, device=self.device, seed=seed)
        images = [load_image_float32(p) for p in tr.image_filenames]
        self.cache = ops.ImageCache.build(images, torch.tensor(tr.metadata["is_thermal"], dtype=torch.float32), torch.arange(len(images)), self.device)
        self.cam_t = {k: tr.cameras[k].to(self.device).contiguous() for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
        self.datamanager = DeviceDataManager(self.cache, self.cam_t, num_rays_per_batch, patch_size)
        self.step = 0
        # mixed_precision=True in thermal-nerfacto's method config (configs/method_configs.py:260): the Trainer runs every iteration under a
        # GradScaler (engine/trainer.py:470-495).  Its device-side equivalent also selects the ONE-call iteration (tn_train_step), which samples the
        # next batch and runs the next iteration's sampling front inside its optimiser launches.
        from .optim import DeviceGradScaler

        self.grad_scaler = DeviceGradScaler(self.device, num_groups=len(self.model.arena.optimised_groups)) if mixed_precision else None

    def train(self, num_steps: int) -> Dict[str, float]:
        """num_steps fused training iterations (callbacks + forward + losses + backward + Adam); returns the last loss dict."""
        self.model.train()
        losses = {}
        if num_steps > 8 and not getattr(self, "_gc_frozen", False):
            # the long-lived heap (torch, the model, the arena views) out of the cyclic collector's full passes: 35-40 ms each, every
            # 50-150 iterations of a 1.3 ms step (the collector stays enabled for what the loop allocates)
            import gc

            gc.collect()
            gc.freeze()
            self._gc_frozen = True
        for _ in range(num_steps):
            o, d, cam, img, is_th = self.datamanager.next_train(self.step)
            rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones_like(o[:, :1]), camera_indices=cam[:, None])
            losses = self.model.train_iteration(rb, {"image": img, "is_thermal": is_th}, self.step, grad_scaler=self.grad_scaler)
            self.step += 1
        return {k: float(v) for k, v in losses.items()}

    def eval_camera(self, i: int):
        c = self.eval_outputs.cameras
        return types.SimpleNamespace(camera_to_worlds=c["c2w"][i], fx=float(c["fx"][i]), fy=float(c["fy"][i]), cx=float(c["cx"][i]), cy=float(c["cy"][i]),
                                     width=int(c["width"][i]), height=int(c["height"][i]), distortion_params=c["distortion"][i], camera_index=i)

    @torch.no_grad()
    def get_average_eval_image_metrics(self) -> Dict[str, float]:
        """pipelines/base_pipeline.py:377-440: every eval image rendered in full, metrics averaged per key."""
        self.model.eval()
        sums: Dict[str, List[float]] = {}
        for i, path in enumerate(self.eval_outputs.image_filenames):
            outs = self.model.get_outputs_for_camera(self.eval_camera(i))
            gt = load_image_float32(path).to(self.device)
            metrics, _ = self.model.get_image_metrics_and_images(outs, {"image": gt, "is_thermal": int(self.eval_outputs.metadata["is_thermal"][i])})
            for k, v in metrics.items():
                sums.setdefault(k, []).append(float(v))
        self.model.train()
        return {k: sum(v) / len(v) for k, v in sums.items()}
