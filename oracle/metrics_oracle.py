"""CPU restatement (numpy / scipy) of the image metric the eval driver reports beside PSNR: SSIM.

TEST INFRASTRUCTURE ONLY (checker for nerfstudio_thermal_amd.model._ssim); never imported by the product path.

PARITY UNPINNED: the reference calls `torchmetrics.functional.structural_similarity_index_measure(gt, pred)` with its defaults
(models/nerfacto.py:247-252,430; models/thermal_nerfacto.py:534,546; `torchmetrics[image]>=1.0.1`, pyproject.toml:58).  torchmetrics is a
third-party package that is absent from /root/reference and from this image, and the reference's own tests hold no SSIM vector, so this file
restates torchmetrics' published algorithm (torchmetrics/functional/image/ssim.py, v1.x: `_ssim_update`):
  gaussian window 11 x 11, sigma 1.5 (separable, normalised); data_range = max(pred.max() - pred.min(), target.max() - target.min()) when not
  given; c1 = (0.01 R)^2, c2 = (0.03 R)^2; both images reflect-padded by (k - 1) / 2; mu, sigma from the windowed first and second moments
  (valid correlation of the padded images = "same" size); ssim map cropped by the padding again; mean over channels and pixels.
It is an INDEPENDENT formulation (explicit double loops of numpy slices instead of a grouped convolution) so that it cross-checks the product
code rather than repeating it.
"""
import numpy as np


def gaussian_1d(kernel_size: int = 11, sigma: float = 1.5) -> np.ndarray:
    ax = np.arange(kernel_size, dtype=np.float64) - (kernel_size - 1) / 2.0
    g = np.exp(-0.5 * (ax / sigma) ** 2)
    return g / g.sum()


def _window_mean(img: np.ndarray, g: np.ndarray) -> np.ndarray:
    """valid 2-D correlation of img [H, W] with the separable window g x g, by shifted slices"""
    k = len(g)
    H, W = img.shape
    rows = np.zeros((H - k + 1, W), dtype=np.float64)
    for i in range(k):
        rows += g[i] * img[i:i + H - k + 1, :]
    out = np.zeros((H - k + 1, W - k + 1), dtype=np.float64)
    for j in range(k):
        out += g[j] * rows[:, j:j + W - k + 1]
    return out


def ssim(pred: np.ndarray, target: np.ndarray, kernel_size: int = 11, sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> float:
    """pred, target: [C, H, W] float arrays -> mean SSIM"""
    pred = np.asarray(pred, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    R = max(pred.max() - pred.min(), target.max() - target.min())
    c1, c2 = (k1 * R) ** 2, (k2 * R) ** 2
    pad = (kernel_size - 1) // 2
    g = gaussian_1d(kernel_size, sigma)
    vals = []
    for c in range(pred.shape[0]):
        p = np.pad(pred[c], pad, mode="reflect")
        t = np.pad(target[c], pad, mode="reflect")
        mu_p, mu_t = _window_mean(p, g), _window_mean(t, g)
        s_pp = _window_mean(p * p, g) - mu_p * mu_p
        s_tt = _window_mean(t * t, g) - mu_t * mu_t
        s_pt = _window_mean(p * t, g) - mu_p * mu_t
        m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
        if m.shape[0] > 2 * pad and m.shape[1] > 2 * pad:
            m = m[pad:-pad, pad:-pad]
        vals.append(m)
    return float(np.mean(np.stack(vals)))
