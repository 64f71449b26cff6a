"""a2: RayBundle / RaySamples / Frustums keep the reference's TensorDataclass semantics (utils/tensor_dataclass.py:27-350,
cameras/rays.py:32-295).  tests/golden/rays.npz holds what the REFERENCE's own classes produce for a fixed script of operations
(oracle/make_golden_rays.py: construction-time broadcasting incl. a dict field and the nested Frustums, int / slice / ellipsis / tensor
indexing, reshape, flatten, broadcast_to, the row-major slice of the chunked eval, get_ray_samples); the same script runs here on this
package's classes and every tensor must come out identical."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import make_golden_rays as gen  # noqa: E402  (only its input builder and operation script: nothing of the reference is imported here)

from nerfstudio_thermal_amd.rays import Frustums, RayBundle, RaySamples  # noqa: E402
from nerfstudio_thermal_amd.tensor_dataclass import TensorDataclass  # noqa: E402


def test_ray_types_match_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "rays.npz"))
    objs, scalars, extra = gen.cases(RayBundle)
    out = {}
    for name, o in objs.items():
        gen.dump(name, o, out)
    ref_keys = sorted(k for k in g.files if not k.startswith(("scalar/", "extra/")))
    assert sorted(out.keys()) == ref_keys
    for k in ref_keys:
        assert out[k].shape == g[k].shape, (k, out[k].shape, g[k].shape)
        assert np.array_equal(out[k], g[k]), k
    for k, v in scalars.items():
        assert int(g["scalar/" + k]) == int(v), k
    for k, v in extra.items():
        assert np.array_equal(v.numpy(), g["extra/" + k]), k


def test_broadcast_views_and_refusals():
    bundle, samples = gen.inputs()
    A = RayBundle(**bundle)
    assert A.shape == (4, 5) and A.directions.shape == (4, 5, 3) and A.pixel_area.shape == (4, 5, 1)
    assert A.directions.stride(0) == 0 and A.pixel_area.stride() == (0, 0, 1)  # broadcast VIEWS, not copies
    assert A.metadata["directions_norm"].shape == (4, 5, 1) and A.camera_indices.shape == (4, 5, 1)
    rs = A.get_ray_samples(**samples)
    assert isinstance(rs, RaySamples) and isinstance(rs.frustums, Frustums) and rs.shape == (4, 5, 7)
    assert rs.frustums.origins.shape == (4, 5, 7, 3) and rs.frustums.origins.stride(2) == 0  # broadcast along the samples
    assert rs.camera_indices.shape == (4, 5, 7, 1) and rs.metadata["directions_norm"].shape == (4, 5, 7, 1)
    with pytest.raises(RuntimeError):
        A[0] = A[1]
    from dataclasses import dataclass

    @dataclass
    class NoTensor(TensorDataclass):
        name: str = "x"

    with pytest.raises(ValueError):
        NoTensor()  # no tensor at all
    # a sampled subset and the truth value of an empty batch
    assert len(A.flatten().sample(7)) == 7
    with pytest.raises(ValueError):
        bool(A.flatten()[0:0])


def test_custom_trailing_dimensions():
    from dataclasses import dataclass

    @dataclass
    class Pose(TensorDataclass):
        c2w: torch.Tensor  # [*batch, 3, 4]
        fx: torch.Tensor  # [*batch, 1]
        _field_custom_dimensions = {"c2w": 2}

    p = Pose(c2w=torch.arange(2 * 3 * 3 * 4, dtype=torch.float32).view(2, 3, 3, 4), fx=torch.ones(3, 1))
    assert p.shape == (2, 3) and p.fx.shape == (2, 3, 1)
    assert p[1].c2w.shape == (3, 3, 4) and p[:, 0].c2w.shape == (2, 3, 4) and p.flatten().c2w.shape == (6, 3, 4)
    assert p.reshape((3, 2)).c2w.shape == (3, 2, 3, 4) and p.broadcast_to((5, 2, 3)).c2w.shape == (5, 2, 3, 3, 4)
    assert torch.equal(p[1, 2].c2w, p.c2w[1, 2])
