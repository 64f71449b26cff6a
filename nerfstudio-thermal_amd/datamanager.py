"""The DataManager seam of SURVEY.md 8b tier 1 over the device-resident pixel sampler (8f N2).

The reference's `VanillaDataManager.next_train` (data/datamanagers/base_datamanager.py:538-547) is per-step HOST work: CacheDataloader hands over
the cached image list, PatchPixelSampler loops over the images in Python (data/pixel_samplers.py:296-337), RayGenerator runs on the indices.
Here the cached images are resident in HBM (ops.ImageCache) and `next_train(step) -> (RayBundle, batch)` is ONE launch (tn_sample_rays: patch
sampling + ground-truth gather + ray generation), with the reference's batch keys (`image`, `indices`, `is_thermal`).

Two classes share `TrainRaySource`:
  * `HipDataManager` / `HipDataManagerConfig` -- stand-alone (this package's dataparser; what bench.py and pipeline.ThermalPipeline drive on a
    box without nerfstudio), with the DataManager interface of base_datamanager.py:113-309: next_train / next_eval / next_eval_image,
    get_train_rays_per_batch / get_eval_rays_per_batch, get_datapath, get_param_groups, get_training_callbacks, train_dataset / eval_dataset
    views (scene_box, metadata, cameras, len, get_image_float32), train_count / eval_count, iter_train / get_train_iterable.
  * `make_nerfstudio_datamanager()` -- inside a nerfstudio installation: a subclass of the reference's own `VanillaDataManager[ThermalDataset]`
    whose `setup_train` / `next_train` are the device path; datasets, dataparser, eval loaders, config fields stay the reference's, so the
    pipeline's isinstance checks (pipelines/base_pipeline.py:362) and `ns-eval` / the viewer see the class they expect.  The method plugin
    installs it as `pipeline.datamanager._target` (plugin.py).
"""
from __future__ import annotations

import random
import types
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Type

import torch
from torch import Tensor

from . import ops
from .data import DeviceDataManager
from .dataparser import DataparserOutputs, ThermalNerfDataParserConfig, load_image_float32
from .rays import RayBundle


def camera_tensors(cameras: Any, device) -> Dict[str, Tensor]:
    """-> {c2w [C,3,4], fx, fy, cx, cy [C], distortion [C,6]} fp32 on `device` from this package's camera dict (dataparser.DataparserOutputs.cameras)
    or from a nerfstudio `Cameras` (cameras/cameras.py:60-140: camera_to_worlds [C,3,4], fx/fy/cx/cy [C,1], distortion_params [C,6] or None,
    camera_type [C,1]).  Only PERSPECTIVE cameras are on the path."""
    dev = torch.device(device)
    if isinstance(cameras, dict):
        src = {k: cameras[k] for k in ("c2w", "fx", "fy", "cx", "cy")}
        dist = cameras.get("distortion")
    else:
        ctype = getattr(cameras, "camera_type", None)
        if ctype is not None and bool((torch.as_tensor(ctype).reshape(-1) != 1).any()):  # CameraType.PERSPECTIVE.value == 1 (cameras/cameras.py:38)
            raise NotImplementedError("the HIP ray generator covers PERSPECTIVE cameras only")
        src = {"c2w": cameras.camera_to_worlds, "fx": cameras.fx, "fy": cameras.fy, "cx": cameras.cx, "cy": cameras.cy}
        dist = getattr(cameras, "distortion_params", None)
    n = int(torch.as_tensor(src["c2w"]).reshape(-1, 3, 4).shape[0])
    out = {"c2w": torch.as_tensor(src["c2w"], dtype=torch.float32).reshape(n, 3, 4).to(dev).contiguous()}
    for k in ("fx", "fy", "cx", "cy"):
        out[k] = torch.as_tensor(src[k], dtype=torch.float32).reshape(n).to(dev).contiguous()
    out["distortion"] = (torch.zeros(n, 6) if dist is None else torch.as_tensor(dist, dtype=torch.float32).reshape(n, 6)).to(dev).contiguous()
    return out


class TrainRaySource:
    """Images of one split resident in HBM + the split's cameras -> `next(step) -> (RayBundle, batch)`.

    batch order = the order CacheDataloader would hand the images to the pixel sampler in (data/utils/dataloaders.py:100-103: a random
    permutation of the dataset drawn once when all images are cached; `shuffle=False`: dataset order)."""

    def __init__(self, images: Sequence[Tensor], is_thermal: Sequence[float], cameras: Any, num_rays: int, patch_size: int, device,
                 ray_bundle_cls: Callable[..., Any] = RayBundle, shuffle: bool = True, prefetch="cowork"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the device datamanager needs a GPU: pixel sampling and ray generation are HIP kernels (there is no CPU fallback)")
        n = len(images)
        order = random.sample(range(n), k=n) if shuffle else list(range(n))
        self.batch_order = order
        th = torch.as_tensor([float(is_thermal[i]) for i in order], dtype=torch.float32)
        self.cache = ops.ImageCache.build([images[i] for i in order], th, torch.tensor(order, dtype=torch.int64), self.device)
        self.cameras = camera_tensors(cameras, self.device)
        self.patch_size = int(patch_size)
        self.num_rays = (int(num_rays) // self.patch_size ** 2) * self.patch_size ** 2  # PatchPixelSampler.set_num_rays_per_batch (data/pixel_samplers.py:380-386)
        if self.num_rays // self.patch_size ** 2 < n:
            raise ValueError(f"{num_rays} rays per batch cannot cover {n} images with {patch_size}x{patch_size} patches")
        # prefetch="cowork" (TRAINING sources): the next batch is handed to the training step in between (ops.sample_rays_deferred: one pending
        # block per process).  Eval sources launch their batches themselves: an eval batch parked in that slot would be sampled by the next TRAINING
        # step and kept alive until the next evaluation, hundreds of iterations later, while the training batches lose their co-work.
        self._dm = DeviceDataManager(self.cache, self.cameras, self.num_rays, self.patch_size, prefetch=prefetch)
        self._bundle = ray_bundle_cls

    def next(self, step: int = 0) -> Tuple[Any, Dict[str, Tensor]]:
        o, d, cam, img, is_th, idx, area, nrm = self._dm.next_train_full(step)
        bundle = self._bundle(origins=o, directions=d, pixel_area=area, camera_indices=cam[:, None], metadata={"directions_norm": nrm})
        return bundle, {"image": img, "indices": idx, "is_thermal": is_th}


# ------------------------------------------------------------------------------------------------ stand-alone (no nerfstudio)
class DatasetView:
    """What the pipeline / trainer read of an InputDataset (data/datasets/base_dataset.py:37-170): scene_box, metadata, cameras, len,
    image_filenames, get_image_float32, __getitem__ -> {"image_idx", "image", "is_thermal"} (ThermalDataset.get_metadata)."""

    exclude_batch_keys_from_device: List[str] = ["image"]

    def __init__(self, outputs: DataparserOutputs):
        from .model import SceneBox

        self._dataparser_outputs = outputs
        self.scene_box = SceneBox(aabb=outputs.scene_box_aabb)
        self.metadata = dict(outputs.metadata)
        self.cameras = outputs.cameras
        self.image_filenames = list(outputs.image_filenames)

    def __len__(self) -> int:
        return len(self.image_filenames)

    def get_image_float32(self, image_idx: int) -> Tensor:
        return load_image_float32(self.image_filenames[image_idx])

    def __getitem__(self, image_idx: int) -> Dict[str, Any]:
        return {"image_idx": image_idx, "image": self.get_image_float32(image_idx), "is_thermal": self.metadata["is_thermal"][image_idx]}

    def camera(self, i: int):
        c = self.cameras
        return types.SimpleNamespace(camera_to_worlds=c["c2w"][i], fx=float(c["fx"][i]), fy=float(c["fy"][i]), cx=float(c["cx"][i]), cy=float(c["cy"][i]),
                                     width=int(c["width"][i]), height=int(c["height"][i]), distortion_params=c["distortion"][i], camera_index=i)


@dataclass
class HipDataManagerConfig:
    """Field names and defaults of VanillaDataManagerConfig as method_configs["thermal-nerfacto"] sets them (configs/method_configs.py:261-268,
    data/datamanagers/base_datamanager.py:309-350)."""

    _target: Type = field(default_factory=lambda: HipDataManager)
    data: Optional[Path] = None
    dataparser: ThermalNerfDataParserConfig = field(default_factory=ThermalNerfDataParserConfig)
    train_num_rays_per_batch: int = 4096 * 2
    train_num_images_to_sample_from: int = -1
    train_num_times_to_repeat_images: int = -1
    train_sample_images_randomly: bool = True
    eval_num_rays_per_batch: int = 4096 * 2
    eval_num_images_to_sample_from: int = -1
    eval_num_times_to_repeat_images: int = -1
    eval_sample_images_randomly: bool = True
    eval_image_indices: Optional[Tuple[int, ...]] = (0,)
    camera_res_scale_factor: float = 1.0
    patch_size: int = 2  # PatchPixelSamplerConfig(patch_size=2) in the method table ("HACK: don't change this, stuff will break")

    def setup(self, **kwargs) -> "HipDataManager":
        return self._target(self, **kwargs)


def _check_supported(config) -> None:
    if int(getattr(config, "train_num_images_to_sample_from", -1)) != -1:
        raise NotImplementedError("the device datamanager caches ALL training images in HBM (train_num_images_to_sample_from must stay -1)")
    if float(getattr(config, "camera_res_scale_factor", 1.0)) != 1.0 and isinstance(config, HipDataManagerConfig):
        raise NotImplementedError("camera_res_scale_factor != 1 needs the reference's dataset (use the nerfstudio datamanager class)")


class HipDataManager(torch.nn.Module):
    """DataManager (data/datamanagers/base_datamanager.py:113-309) on this package's dataparser; see the module docstring."""

    includes_time = False
    train_sampler = None
    eval_sampler = None

    def __init__(self, config: HipDataManagerConfig, device="cuda", test_mode: str = "val", world_size: int = 1, local_rank: int = 0, **kwargs):
        super().__init__()
        _check_supported(config)
        self.config, self.device, self.world_size, self.local_rank, self.test_mode = config, torch.device(device), world_size, local_rank, test_mode
        if config.data is not None:
            config.dataparser.data = str(config.data)
        self.dataparser = config.dataparser.setup()
        self.train_dataparser_outputs: DataparserOutputs = self.dataparser.get_dataparser_outputs("train")
        self.train_dataset = DatasetView(self.train_dataparser_outputs)
        self.eval_dataset = DatasetView(self.dataparser.get_dataparser_outputs("test" if test_mode in ("test", "inference") else "val"))
        self.train_count = self.eval_count = 0
        self._train_src: Optional[TrainRaySource] = None
        self._eval_src: Optional[TrainRaySource] = None
        if test_mode != "inference":
            self.setup_train()
            self.setup_eval()

    def forward(self):
        raise NotImplementedError

    @staticmethod
    def _source(ds: DatasetView, num_rays: int, patch: int, device, shuffle: bool, prefetch="cowork") -> TrainRaySource:
        return TrainRaySource([ds.get_image_float32(i) for i in range(len(ds))], ds.metadata["is_thermal"], ds.cameras, num_rays, patch, device,
                              shuffle=shuffle, prefetch=prefetch)

    def setup_train(self) -> None:
        self._train_src = self._source(self.train_dataset, self.config.train_num_rays_per_batch, self.config.patch_size, self.device,
                                       self.config.train_sample_images_randomly)

    def setup_eval(self) -> None:
        self._eval_next_image = 0
        if len(self.eval_dataset) and self.config.eval_num_rays_per_batch // self.config.patch_size ** 2 >= len(self.eval_dataset):
            self._eval_src = self._source(self.eval_dataset, self.config.eval_num_rays_per_batch, self.config.patch_size, self.device,
                                          self.config.eval_sample_images_randomly, prefetch=False)

    def iter_train(self) -> None:
        self.train_count = 0

    def iter_eval(self) -> None:
        self.eval_count = 0

    def get_train_iterable(self, length: int = -1):
        self.iter_train()
        i = 0
        while length < 0 or i < length:
            yield self.next_train(i)
            i += 1

    def next_train(self, step: int) -> Tuple[RayBundle, Dict[str, Tensor]]:
        self.train_count += 1
        return self._train_src.next(step)

    def next_eval(self, step: int) -> Tuple[RayBundle, Dict[str, Tensor]]:
        if self._eval_src is None:
            raise ValueError("no eval images (or fewer eval rays per batch than eval images)")
        self.eval_count += 1
        return self._eval_src.next(step)

    def next_eval_image(self, step: int):
        """-> (camera, batch) of one eval image, round robin (the reference draws them at random: RandIndicesEvalDataloader)."""
        if not len(self.eval_dataset):
            raise ValueError("No more eval images")
        i = self._eval_next_image % len(self.eval_dataset)
        self._eval_next_image += 1
        return self.eval_dataset.camera(i), self._image_batch(i)

    def _image_batch(self, i: int) -> Dict[str, Any]:
        return {"image_idx": i, "image": self.eval_dataset.get_image_float32(i).to(self.device), "is_thermal": self.eval_dataset.metadata["is_thermal"][i]}

    @property
    def fixed_indices_eval_dataloader(self):
        """every eval image once, in order (FixedIndicesEvalDataloader, data/utils/dataloaders.py:201-246)"""
        return [(self.eval_dataset.camera(i), self._image_batch(i)) for i in range(len(self.eval_dataset))]

    def get_train_rays_per_batch(self) -> int:
        return self._train_src.num_rays if self._train_src is not None else self.config.train_num_rays_per_batch

    def get_eval_rays_per_batch(self) -> int:
        return self._eval_src.num_rays if self._eval_src is not None else self.config.eval_num_rays_per_batch

    def get_datapath(self) -> Path:
        return Path(self.config.dataparser.data)

    def get_training_callbacks(self, training_callback_attributes=None) -> list:
        return []

    def get_param_groups(self) -> Dict[str, list]:
        return {}


# ------------------------------------------------------------------------------------------------ inside nerfstudio
def make_nerfstudio_datamanager():
    """-> the class `HipVanillaDataManager(VanillaDataManager[ThermalDataset])`: the reference's datamanager with its TRAIN side on the device.
    Raises ImportError when nerfstudio is not importable (the caller -- plugin.py -- decides what that means)."""
    from nerfstudio.cameras.rays import RayBundle as NsRayBundle
    from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManager
    from nerfstudio.data.datasets.thermal_dataset import ThermalDataset

    class HipVanillaDataManager(VanillaDataManager[ThermalDataset]):  # type: ignore[misc]
        """VanillaDataManager whose setup_train / next_train (data/datamanagers/base_datamanager.py:491-509,538-547) keep the training images in
        HBM and draw a batch in one launch.  Everything else -- dataparser, ThermalDataset, the eval loaders, the config -- is the reference's."""

        def setup_train(self):
            assert self.train_dataset is not None
            _check_supported(self.config)
            ds = self.train_dataset
            if getattr(ds, "_dataparser_outputs", None) is not None and getattr(ds._dataparser_outputs, "mask_filenames", None):
                raise NotImplementedError("masked pixel sampling is outside the thermal-nerfacto path")
            ps = getattr(self.config.pixel_sampler, "patch_size", None) or self.config.patch_size
            is_th = ds.metadata["is_thermal"]
            self._train_src = TrainRaySource([ds.get_image_float32(i) for i in range(len(ds))], [float(is_th[i]) for i in range(len(ds))], ds.cameras,
                                             self.config.train_num_rays_per_batch, ps, self.device, ray_bundle_cls=NsRayBundle,
                                             shuffle=self.config.train_sample_images_randomly)
            # what trainer / callbacks read of the sampler (engine/trainer.py:263; models read nothing of it)
            src = self._train_src

            def _set(n, src=src, ps=ps):
                raise NotImplementedError("the batch size of the device sampler is fixed at setup")

            self.train_pixel_sampler = types.SimpleNamespace(num_rays_per_batch=src.num_rays, set_num_rays_per_batch=_set, config=self.config.pixel_sampler)

        def next_train(self, step: int):
            self.train_count += 1
            return self._train_src.next(step)

        def get_train_rays_per_batch(self) -> int:
            return self._train_src.num_rays

    HipVanillaDataManager.__qualname__ = "HipVanillaDataManager"
    return HipVanillaDataManager
