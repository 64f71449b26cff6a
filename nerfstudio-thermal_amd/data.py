"""Device-resident data manager for the training loop: what VanillaDataManager.next_train does on the host every step
(data/datamanagers/base_datamanager.py:538-547: PatchPixelSampler over the cached images -> ray indices + ground truth -> RayGenerator), done
on the GPU.  `prefetch="cowork"` (the default) hands the NEXT batch's sampling to the training step that runs in between: tn_train_step samples it
in co-work blocks of its optimiser launch (ops.sample_rays_deferred / TnTrainStep.next_sample), 7 us less at the head of every iteration; when no
such step comes first (any other training path, evaluation) the batch is launched as before, just before it is handed out.  The uniforms of
batch k + 1 are drawn when batch k is handed out: the same sequence of draws, one step earlier.  `prefetch=True` prepares the next batch on a
side stream instead; on one MI355X that measured SLOWER than the launch in line (1.593 vs 1.565 ms/step: the extra stream and event traffic cost
more than they hide).  `prefetch=False`: one launch per call."""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops


class DeviceDataManager:
    def __init__(self, cache: "ops.ImageCache", cameras: Dict[str, Tensor], num_rays: int, patch_size: int = 2, prefetch="cowork",
                 side_stream: Optional[torch.cuda.Stream] = None):
        """cameras: c2w [C,3,4], fx, fy, cx, cy [C], distortion [C,6] on the device, indexed by the dataset (camera) index."""
        self.cache, self.cam, self.num_rays, self.patch = cache, cameras, int(num_rays), int(patch_size)
        self.device = cache.buffer.device
        if prefetch == "cowork" and os.environ.get("TN_DM_PREFETCH", "1") == "0":  # (A/B timing: one launch per call)
            prefetch = False
        self.prefetch = prefetch
        self._side = side_stream
        self._pending: Optional[Tuple[Tuple[Tensor, ...], torch.cuda.Event]] = None
        self._rand = ops.UniformPool(self.device)
        self._ready: Optional[Tuple[Tensor, ...]] = None  # prefetch="cowork": the batch handed to the training step in between (8 tensors)

    def _next_cowork(self) -> Tuple[Tensor, ...]:
        n = self.num_rays
        if self._ready is None:  # first call: this batch in line
            u = self._rand.take((n // (self.patch * self.patch), 3))
            batch = ops.sample_rays(self.cache, n, u, self.cam, self.patch, with_bundle_extras=True)
        else:
            ops.flush_pending_sample()  # (a no-op when a tn_train_step has sampled it)
            batch = self._ready
        u = self._rand.take((n // (self.patch * self.patch), 3))
        self._ready = ops.sample_rays_deferred(self.cache, n, u, self.cam, self.patch)
        return batch

    def next_train_full(self, step: int = 0) -> Tuple[Tensor, ...]:
        """-> origins, directions, camera_indices, image, is_thermal, ray_indices [N,3] int64 (camera,row,col), pixel_area [N,1], directions_norm [N,1]:
        every field of the reference's (RayBundle, batch) pair (datamanager.TrainRaySource builds the pair).  Same launch as next_train."""
        if self.prefetch == "cowork":
            return self._next_cowork()
        n = self.num_rays
        u = self._rand.take((n // (self.patch * self.patch), 3))
        return ops.sample_rays(self.cache, n, u, self.cam, self.patch, with_bundle_extras=True)

    def _make(self) -> Tuple[Tensor, ...]:
        n = self.num_rays
        u = self._rand.take((n // (self.patch * self.patch), 3))  # what PatchPixelSampler draws with torch.rand (drawn 32 steps at a time)
        # pixel sampler + GT gather + raygen: one launch.  (pixel_area is computed as the reference's RayGenerator computes it every iteration,
        # although thermal-nerfacto never reads it; ops.sample_rays(want_pixel_area=False) would skip two of the three undistortions per ray.)
        o, d, cam, img, is_th, _ = ops.sample_rays(self.cache, n, u, self.cam, self.patch)
        return o, d, cam, img, is_th

    def _launch_prefetch(self) -> None:
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)  # keeps the RNG draws of consecutive batches in program order
        with torch.cuda.stream(self._side):
            batch = self._make()
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._pending = (batch, ev)

    def next_train(self, step: int = 0) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
        """-> origins [N,3], directions [N,3], camera_indices [N] int64, image [N,3], is_thermal [N] of a fresh pixel batch."""
        if self.prefetch == "cowork":
            return self._next_cowork()[:5]
        if not self.prefetch:
            return self._make()
        if self._pending is None:
            self._launch_prefetch()
        batch, ev = self._pending
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        for t in batch:
            t.record_stream(main)  # allocated on the side stream, consumed on this one
        self._launch_prefetch()  # the batch after this one, while this step computes
        return batch
