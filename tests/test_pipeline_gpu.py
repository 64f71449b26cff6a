"""N3 end to end from DISK: the synthetic cube scene written as an RGB+T transforms.json dataset, parsed, trained with the fused step and
evaluated image by image (pipeline.ThermalPipeline: pipelines/base_pipeline.py:230-470 for this method)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_train_and_eval_from_disk(tmp_path):
    import train_eval_scene as T
    from nerfstudio_thermal_amd.pipeline import ThermalPipeline

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    path = T.write_cube_scene(str(tmp_path), 10, dev)
    assert os.path.exists(path) and len(os.listdir(tmp_path / "images")) == 10 and len(os.listdir(tmp_path / "images_thermal")) == 10
    pipe = ThermalPipeline(str(tmp_path), device=dev)
    assert len(pipe.train_outputs.image_filenames) == 18 and len(pipe.eval_outputs.image_filenames) == 2
    assert pipe.train_outputs.metadata["is_thermal"] == [0] * 9 + [1] * 9 and pipe.eval_outputs.metadata["is_thermal"] == [0, 1]
    first = pipe.train(50)
    last = pipe.train(950)
    assert all(np.isfinite(v) for v in last.values())
    assert last["rgb_loss"] < 0.2 * first["rgb_loss"] and last["thermal_loss"] < 0.2 * first["thermal_loss"], (first, last)
    m = pipe.get_average_eval_image_metrics()
    assert set(m) == {"psnr_rgb", "ssim_rgb", "psnr_thermal", "ssim_thermal"} and all(np.isfinite(v) for v in m.values())
    # held-out views of a consistent scene: the structure is there after 1000 iterations (the absolute PSNR is limited by the reference's
    # averaged appearance embedding on an 18-image scene)
    assert m["ssim_rgb"] > 0.5 and m["ssim_thermal"] > 0.4 and m["psnr_rgb"] > 9.0 and m["psnr_thermal"] > 9.0, m
