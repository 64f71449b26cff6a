"""Golden vectors for the boundary TYPES: the reference's own RayBundle / RaySamples / Frustums (cameras/rays.py over
utils/tensor_dataclass.py) driven through construction-time broadcasting, indexing, reshape / flatten / broadcast_to, the row-major slice
used by the chunked eval and get_ray_samples.  Build container only; writes tests/golden/rays.npz (arrays only).

    python oracle/make_golden_rays.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def inputs():
    """Seeded inputs shared with tests/test_rays_cpu.py (nerfstudio-thermal_amd/synth.py integer-hash generator)."""
    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import synth

    u = lambda name, shape, lo=-1.0, hi=1.0: torch.from_numpy(synth.uniform("rays_" + name, shape, lo, hi, 0))  # noqa: E731
    bundle = dict(origins=u("o", (4, 5, 3)), directions=u("d", (5, 3)), pixel_area=u("a", (1, 1), 0.1, 1.0),
                  camera_indices=torch.arange(4).view(4, 1, 1), metadata={"directions_norm": u("n", (4, 5, 1), 0.5, 1.5)}, times=u("t", (4, 1, 1), 0.0, 1.0))
    edges = torch.sort(u("bins", (4, 5, 8, 1), 0.05, 6.0), dim=-2).values
    sedges = torch.sort(u("sbins", (4, 5, 8, 1), 0.0, 1.0), dim=-2).values
    samples = dict(bin_starts=edges[..., :-1, :], bin_ends=edges[..., 1:, :], spacing_starts=sedges[..., :-1, :], spacing_ends=sedges[..., 1:, :])
    return bundle, samples


def dump(prefix, obj, out):
    """Every tensor of a (nested) tensor dataclass / dict under `prefix/...`, plus its batch shape."""
    import dataclasses

    if hasattr(obj, "shape") and dataclasses.is_dataclass(obj):
        out[prefix + "/__shape__"] = np.asarray(tuple(obj.shape), dtype=np.int64)
        for f in dataclasses.fields(obj):
            v = getattr(obj, f.name)
            if isinstance(v, torch.Tensor):
                out[f"{prefix}/{f.name}"] = v.detach().numpy()
            elif isinstance(v, dict):
                for k, x in v.items():
                    if isinstance(x, torch.Tensor):
                        out[f"{prefix}/{f.name}.{k}"] = x.detach().numpy()
            elif dataclasses.is_dataclass(v):
                dump(f"{prefix}/{f.name}", v, out)


def cases(RayBundle):
    """name -> object, built with whichever RayBundle class is passed (the reference's here, this package's in the test)."""
    bundle, samples = inputs()
    A = RayBundle(**bundle)
    idx = torch.tensor([[0, 1], [3, 2]])
    rs = A.get_ray_samples(**samples)
    out = {
        "A": A, "A[...,0]": A[..., 0], "A[1:3,::2]": A[1:3, ::2], "A[2]": A[2], "A[idx]": A[idx], "A.flatten": A.flatten(),
        "A.reshape(2,10)": A.reshape((2, 10)), "A.rowmajor(3,11)": A.get_row_major_sliced_ray_bundle(3, 11), "A.broadcast(2,4,5)": A.broadcast_to((2, 4, 5)),
        "RS": rs, "RS[...,2]": rs[..., 2], "RS[1]": rs[1], "RS.flatten": rs.flatten(), "RS.reshape(20,7)": rs.reshape((20, 7)),
    }
    scalars = {"len(A)": len(A), "A.size": A.size, "A.ndim": A.ndim, "len(A.flatten)": len(A.flatten()), "RS.size": rs.size, "RS.ndim": rs.ndim,
               "len(RS)": len(rs)}
    extra = {"RS.positions": rs.frustums.get_positions(), "RS.start_positions": rs.frustums.get_start_positions()}
    return out, scalars, extra


def main():
    import ref_import

    ref_import.import_reference()
    from nerfstudio.cameras.rays import RayBundle

    objs, scalars, extra = cases(RayBundle)
    out = {}
    for name, o in objs.items():
        dump(name, o, out)
    for k, v in scalars.items():
        out["scalar/" + k] = np.int64(v)
    for k, v in extra.items():
        out["extra/" + k] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", "rays.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), len(out), "arrays")


if __name__ == "__main__":
    main()
