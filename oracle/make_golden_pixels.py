"""Golden vectors for the on-device pixel sampler, produced by the REFERENCE's PatchPixelSampler on a jagged RGB + thermal
image list (run in the build container only; /root/reference is imported through oracle/ref_import.py).

Inputs are regenerated on both sides from nerfstudio_thermal_amd.synth (cameras, images, uniforms); only the reference's outputs are stored:
    tests/golden/pixels.npz : indices [N,3], image [N,3], is_thermal [N], plus the batch order (image_idx) used.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402

ref_import.install_stubs()
sys.path.insert(0, ref_import.REFERENCE_ROOT)

from nerfstudio.data.pixel_samplers import PatchPixelSampler, PatchPixelSamplerConfig  # noqa: E402

import nerfstudio_thermal_amd  # noqa: E402,F401
from nerfstudio_thermal_amd import synth  # noqa: E402
from make_golden import _InjectRand  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
N_RAYS = 256
# batch position -> dataset (camera) index: CacheDataloader hands the images over in sampled order, not in dataset order
BATCH_ORDER = [5, 0, 7, 2, 1, 6, 3, 4]


def batch_inputs():
    cams = synth.synth_cameras()
    imgs = synth.synth_images(cams)
    order = np.array(BATCH_ORDER, dtype=np.int64)
    images = [torch.from_numpy(imgs[c]) for c in order]
    is_thermal = torch.from_numpy(cams["is_thermal"][order].astype(np.float32))
    return cams, images, is_thermal, torch.from_numpy(order)


def main():
    cams, images, is_thermal, image_idx = batch_inputs()
    per_image = (N_RAYS // len(images)) // 4
    u = torch.from_numpy(synth.synth_patch_uniforms(per_image * len(images)))
    sampler = PatchPixelSampler(PatchPixelSamplerConfig(patch_size=2, num_rays_per_batch=N_RAYS))
    batch = {"image": images, "image_idx": image_idx, "is_thermal": is_thermal}
    with _InjectRand([u[i * per_image:(i + 1) * per_image] for i in range(len(images))]):
        out = sampler.sample(batch)
    np.savez_compressed(os.path.join(GOLDEN, "pixels.npz"), num_rays=N_RAYS, batch_order=np.array(BATCH_ORDER),
                        indices=out["indices"].numpy(), image=out["image"].numpy(), is_thermal=out["is_thermal"].numpy())
    print("pixels.npz", out["indices"].shape, out["image"].shape, out["is_thermal"].shape, out["indices"][:6].tolist())


if __name__ == "__main__":
    main()
