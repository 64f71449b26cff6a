"""torch.autograd wrappers of the loss kernels: what lets ThermalNerfactoModel.get_metrics_dict / get_loss_dict
(models/thermal_nerfacto.py:253-388) run on the HIP kernels while the reference Trainer still calls
`functools.reduce(torch.add, loss_dict.values()).backward()` (engine/trainer.py:483-487).

Every loss kernel produces the value AND the gradient in one pass (csrc/tn_misc.hip, tn_sampler.hip), so each Function runs its kernel in
forward, keeps the gradient, and scales it by the incoming scalar gradient in backward: no recomputation, no O(S^2) tape for the distortion
loss.  Arithmetic lives in libthermal_nerf_hip.so; nothing here has a CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops


def _scalar_grads_equal(gs) -> Optional[Tensor]:
    """The common value if all incoming scalar gradients are the same tensor value (the Trainer sums the loss dict: all ones), else None."""
    gs = [g for g in gs if g is not None]
    if not gs:
        return None
    g0 = gs[0]
    for g in gs[1:]:
        if g is not g0 and not bool(torch.equal(g, g0)):
            return None
    return g0


class PixelLosses(torch.autograd.Function):
    """rgb MSE, thermal MSE x thermal_mult, 2x2-patch TV x tv_mult, cross-channel x cross_mult (tn_pixel_losses) -> four scalars + the number
    of RGB rays of the batch (not differentiable; the PSNR metrics are derived from it).  pred_rgb [N,3] and pred_thermal [N,1] may be
    views of one [N,4] buffer (shared mode)."""

    @staticmethod
    def forward(ctx, pred_rgb: Tensor, pred_thermal: Tensor, image: Tensor, is_thermal: Tensor, thermal_mult: float, tv_mult: float,
                cross_mult: float) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
        N = pred_rgb.shape[0]
        shared = (pred_rgb.stride(0) == 4 and pred_thermal.stride(0) == 4 and pred_thermal.data_ptr() == pred_rgb.data_ptr() + 12)
        if shared:
            g = torch.zeros((N, 4), device=pred_rgb.device)
            d_rgb, d_th = g[:, :3], g[:, 3:]
        else:
            pred_rgb, pred_thermal = pred_rgb.contiguous(), pred_thermal.contiguous()
            d_rgb, d_th = torch.zeros_like(pred_rgb), torch.zeros_like(pred_thermal)
        L = torch.zeros(8, device=pred_rgb.device)
        ops.pixel_losses(pred_rgb, pred_thermal, image.contiguous(), is_thermal.contiguous(), thermal_mult, tv_mult, cross_mult, L, d_rgb, d_th)
        ctx.save_for_backward(d_rgb, d_th)
        n_rgb = L[4]
        ctx.mark_non_differentiable(n_rgb)
        return L[0], L[1], L[2], L[3], n_rgb

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, _g4):
        d_rgb, d_th = ctx.saved_tensors
        g = _scalar_grads_equal((g0, g1, g2, g3))
        if g is None:
            raise RuntimeError("PixelLosses: the four pixel-loss terms must enter the total with the same weight (the Trainer sums the loss dict); "
                               "scale a term through its *_loss_mult instead")
        return d_rgb * g, d_th * g, None, None, None, None, None


class DistortionLoss(torch.autograd.Function):
    """distortion_loss (model_components/losses.py:139-158) of one branch: weights [N,S,1], s_bins [N,S+1] -> scalar (mean over rays)."""

    @staticmethod
    def forward(ctx, weights: Tensor, s_bins: Tensor) -> Tensor:
        w = weights[..., 0].contiguous()
        L = torch.zeros(1, device=w.device)
        dw = torch.zeros_like(w)
        ops.distortion_loss(s_bins, w, 1.0, L, dw)
        ctx.save_for_backward(dw)
        return L[0]

    @staticmethod
    def backward(ctx, g):
        (dw,) = ctx.saved_tensors
        return (dw * g).unsqueeze(-1), None


class InterlevelLoss(torch.autograd.Function):
    """One term of interlevel_loss (model_components/losses.py:117-135): proposal weights [N,Sp,1] against the (detached) fine level."""

    @staticmethod
    def forward(ctx, w_prop: Tensor, s_prop: Tensor, w_fine: Tensor, s_fine: Tensor) -> Tensor:
        wp = w_prop[..., 0].contiguous()
        L = torch.zeros(1, device=wp.device)
        dw = torch.zeros_like(wp)
        ops.interlevel_loss(s_fine, w_fine[..., 0].contiguous(), s_prop, wp, 1.0, L, dw)
        ctx.save_for_backward(dw)
        return L[0]

    @staticmethod
    def backward(ctx, g):
        (dw,) = ctx.saved_tensors
        return (dw * g).unsqueeze(-1), None, None, None


class AsymmetricL1(torch.autograd.Function):
    """(gx + gy) * mean|x - y| whose gradient reaches x with weight gx and y with weight gy: the detach pattern of the density loss
    (models/thermal_nerfacto.py:328-344: a * L1(x.detach(), y) + b * L1(x, y.detach()) -> gx = b, gy = a)."""

    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor, gx: float, gy: float) -> Tensor:
        xc, yc = x.contiguous(), y.contiguous()
        L = torch.zeros(1, device=x.device)
        dx, dy = torch.zeros_like(xc), torch.zeros_like(yc)
        ops.l1_loss(xc, yc, gx, gy, L, dx, dy)
        ctx.save_for_backward(dx, dy)
        return L[0]

    @staticmethod
    def backward(ctx, g):
        dx, dy = ctx.saved_tensors
        return dx * g, dy * g, None, None


class CameraRegularizer(torch.autograd.Function):
    """CameraOptimizer.get_loss_dict (cameras/camera_optimizers.py:189-195) through tn_camera_reg."""

    @staticmethod
    def forward(ctx, pose: Tensor, trans_pen: float, rot_pen: float, scale: float) -> Tensor:
        L = torch.zeros(1, device=pose.device)
        gp = torch.zeros_like(pose)
        ops.camera_reg(pose.detach().contiguous(), trans_pen, rot_pen, scale, L, gp)
        ctx.save_for_backward(gp)
        return L[0]

    @staticmethod
    def backward(ctx, g):
        (gp,) = ctx.saved_tensors
        return gp * g, None, None, None
