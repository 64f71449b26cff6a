"""model._ssim (what ThermalNerfactoModel.get_image_metrics_and_images reports as ssim_rgb / ssim_thermal, models/thermal_nerfacto.py:534,546)
against oracle/metrics_oracle.py, an independent numpy restatement of torchmetrics' published SSIM (parity unpinned: torchmetrics is absent),
plus known answers."""
import numpy as np
import torch

import metrics_oracle as mo
import nerfstudio_thermal_amd  # noqa: F401
from nerfstudio_thermal_amd.model import _psnr, _ssim


def test_ssim_matches_independent_restatement():
    g = np.random.default_rng(0)
    for C, H, W in ((3, 48, 64), (1, 30, 41), (3, 120, 160)):
        t = g.uniform(0, 1, (C, H, W)).astype(np.float32)
        # a correlated prediction (blur + noise + gain), as a render is: SSIM well inside (0, 1)
        p = np.clip(0.8 * t + 0.1 * np.roll(t, 1, axis=2) + 0.05 * g.standard_normal((C, H, W)).astype(np.float32), 0, 1)
        got = float(_ssim(torch.from_numpy(p)[None], torch.from_numpy(t)[None]))
        ref = mo.ssim(p, t)
        assert 0.05 < ref < 0.99
        assert abs(got - ref) <= 2e-5, (C, H, W, got, ref)


def test_ssim_known_answers():
    g = np.random.default_rng(1)
    x = g.uniform(0, 1, (3, 40, 40)).astype(np.float32)
    xt = torch.from_numpy(x)[None]
    assert abs(float(_ssim(xt, xt)) - 1.0) < 1e-6 and abs(mo.ssim(x, x) - 1.0) < 1e-12
    # a constant image against itself plus a constant offset d: sigma terms vanish, ssim = (2 a (a + d) + c1) / (a^2 + (a + d)^2 + c1), c1 from
    # the data range = 0 -> c1 = c2 = 0 would be 0/0 on the sigma factor; use a ramp instead: ssim(x, 1 - x) has negative structure term
    r = np.tile(np.linspace(0, 1, 40, dtype=np.float32), (40, 1))[None]
    assert mo.ssim(r, 1 - r) < 0 and float(_ssim(torch.from_numpy(r)[None], torch.from_numpy(1 - r)[None])) < 0
    assert abs(mo.ssim(r, 1 - r) - float(_ssim(torch.from_numpy(r)[None], torch.from_numpy(1 - r)[None]))) < 2e-5
    # PSNR(data_range 1): an error of exactly 0.1 everywhere is 20 dB
    assert abs(float(_psnr(xt, xt + 0.1)) - 20.0) < 1e-4
