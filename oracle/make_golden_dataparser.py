"""Golden vectors for the RGB+T dataparser: a synthetic 10 + 10 frame dataset is written in the on-disk format of
process_data/rgbt_to_nerfstudio_dataset.py (by nerfstudio-thermal_amd/dataparser.write_rgbt_dataset) and parsed by the REFERENCE's own
ThermalNerf dataparser (data/dataparsers/thermalnerf_dataparser.py) for the train and val splits.  Build container only; writes
tests/golden/dataparser.npz (arrays + relative file names).

    python oracle/make_golden_dataparser.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def dataset_inputs(num_per_spectrum=10):
    """Cameras (tilted and shifted so that the 'up' orientation and the centring have something to do) and small synthetic images."""
    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import synth

    cams = synth.synth_cameras(num_per_spectrum, num_per_spectrum)
    a = np.deg2rad(20.0)
    R = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]], dtype=np.float32)
    c2w = cams["c2w"].copy()
    c2w[:, :, :3] = R @ c2w[:, :, :3]
    c2w[:, :, 3] = c2w[:, :, 3] @ R.T * 2.5 + np.array([0.3, -0.2, 0.1], dtype=np.float32)
    cams["c2w"] = c2w.astype(np.float32)
    images = synth.synth_images(cams)
    return cams, images


def main():
    import ref_import

    ref_import.import_reference()
    from pathlib import Path

    from nerfstudio.data.dataparsers.thermalnerf_dataparser import ThermalNerfDataParserConfig
    from nerfstudio_thermal_amd.dataparser import write_rgbt_dataset

    cams, images = dataset_inputs()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        write_rgbt_dataset(d, cams, images)
        for split in ("train", "val"):
            o = ThermalNerfDataParserConfig(data=Path(d), downscale_factor=1).setup().get_dataparser_outputs(split=split)
            c = o.cameras
            out[f"{split}/filenames"] = np.array([os.path.relpath(str(p), d) for p in o.image_filenames])
            out[f"{split}/c2w"] = c.camera_to_worlds.numpy()
            for k, v in (("fx", c.fx), ("fy", c.fy), ("cx", c.cx), ("cy", c.cy), ("width", c.width), ("height", c.height)):
                out[f"{split}/{k}"] = v.reshape(-1).numpy()
            out[f"{split}/distortion"] = c.distortion_params.numpy()
            out[f"{split}/is_thermal"] = np.asarray(o.metadata["is_thermal"], dtype=np.int64)
            out[f"{split}/scale"] = np.float64(o.dataparser_scale)
            out[f"{split}/transform"] = o.dataparser_transform.numpy()
            out[f"{split}/aabb"] = o.scene_box.aabb.numpy()
    path = os.path.join(ROOT, "tests", "golden", "dataparser.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), {k: v.shape for k, v in out.items() if k.startswith("train/")})


if __name__ == "__main__":
    main()
