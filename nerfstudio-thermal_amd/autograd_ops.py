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


def _scalar_grads_equal(gs, who: str = "loss node") -> Optional[Tensor]:
    """The common incoming gradient of the loss terms `gs` (the Trainer sums the loss dict, engine/trainer.py:483: every term arrives with the
    same tensor).  The kernels accumulate the gradients of ALL terms into shared buffers, so the terms cannot be weighted one by one:
      * every term None (nothing of this node is being differentiated) -> None;
      * a term with NO incoming gradient beside terms that have one (e.g. `loss_dict["rgb_loss"].backward()`, or a sum over a subset) has
        weight 0, which differs from the others' weight -> refused loudly, never answered with the gradient of all terms;
      * different values -> refused as well."""
    if all(g is None for g in gs):
        return None
    if any(g is None for g in gs):
        raise RuntimeError(f"{who}: backward reached only a subset of the loss terms (the others have no incoming gradient, i.e. weight 0). The HIP loss "
                           "kernels produce the gradient of the SUM of all terms in one pass; backpropagate the sum of the whole loss dict as the "
                           "reference Trainer does (engine/trainer.py:483), and switch a term off through its *_loss_mult")
    g0 = gs[0]
    for g in gs[1:]:
        if g is g0 or (g.data_ptr() == g0.data_ptr() and g.shape == g0.shape):
            continue  # one tensor handed down a chain of additions: no device round trip
        if not bool(torch.equal(g, g0)):
            raise RuntimeError(f"{who}: the loss terms must enter the total with one common weight (the Trainer sums the loss dict); scale a term "
                               "through its *_loss_mult instead")
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
        g = _scalar_grads_equal((g0, g1, g2, g3), "PixelLosses")
        if g is None:
            return None, None, None, None, None, None, None
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


class LossSpec:
    """What TrainLosses needs besides tensors: the loss multipliers of the config and, per branch, the (non-differentiable) s-space bins of
    the three levels and whether the proposal networks take a gradient this iteration."""

    def __init__(self, separate: bool, thermal_mult: float, tv_mult: float, cross_mult: float, distortion_mult: float, interlevel_mult: float,
                 branches, density_weights=None, camera_regs=()):
        self.separate = separate
        self.thermal_mult, self.tv_mult, self.cross_mult = thermal_mult, tv_mult, cross_mult
        self.distortion_mult, self.interlevel_mult = distortion_mult, interlevel_mult
        self.branches = branches  # [(suffix, [s_bins0, s_bins1, s_bins2], prop_grad)]
        self.density_weights = density_weights  # (a, b) of the density loss or None
        self.camera_regs = camera_regs  # [(trans_pen, rot_pen, scale)] per pose parameter passed


_LOSS_LAYOUTS: dict = {}


class TrainLosses(torch.autograd.Function):
    """EVERY loss term of a training iteration (models/thermal_nerfacto.py:253-388) as one autograd node, through the same launches as the
    fused step (engine.RenderEngine.loss_and_backward): tn_train_losses per branch (distortion + both interlevel terms, the pixel terms in the
    first one), tn_l1_loss x2 (density cross terms), tn_losses_finish (sums + camera regulariser), tn_camera_reg for a second pose tensor.
    Every kernel yields value AND gradient; all gradient buffers come out of one zero-filled allocation, and backward scales that allocation once by the incoming gradient -- the Trainer sums the loss dict
    (engine/trainer.py:483), so all terms arrive with the same weight; anything else is refused (use the *_loss_mult settings).

    inputs : spec, image [N,3], is_thermal [N], then per branch (comp, w0, w1, w2), then (density2, density_thermal, density,
             density2_thermal) when the density loss is on, then one pose parameter per entry of spec.camera_regs.
    outputs: L[0..31] unbound: 0 rgb 1 thermal 2 tv 3 cross 4 #rgb rays 5 #thermal rays 8 interlevel 9 distortion 10 density 11/12 camera
             regularisers; metrics (not differentiable, tn_train_metrics): 16 psnr_rgb 17 psnr_thermal 18/19 |pose[:, :3]|, |pose[:, 3:]| of the
             first pose parameter, 20/21 of the second."""

    @staticmethod
    def forward(ctx, spec: LossSpec, image: Tensor, is_thermal: Tensor, *ts: Tensor):
        dev = image.device
        it = iter(ts)
        per = [(sfx, sb, pg, next(it), [next(it), next(it), next(it)]) for sfx, sb, pg in spec.branches]
        dens = [next(it) for _ in range(4)] if spec.density_weights is not None else []
        poses = [next(it) for _ in spec.camera_regs]
        # one zero-filled allocation for the loss vector and every gradient buffer (order = input order)
        shapes = [(32,), (ops.LOSS_LINES, 16)]  # the loss vector, then the lines the loss launches spread their sums over (ops.train_losses)
        for _, _, pg, comp, ws in per:
            shapes.append(tuple(comp.shape))
            shapes += [tuple(w.shape[:2]) for w in ws]
        shapes += [tuple(d.shape) for d in dens] + [tuple(p.shape) for p in poses]
        lay = _LOSS_LAYOUTS.get(tuple(shapes))
        if lay is None:  # (the shapes of an iteration never change: offsets / strides computed once)
            sizes = [int(torch.Size(s).numel()) for s in shapes]
            offs, tot = [], 0
            for n in sizes:
                offs.append(tot)
                tot += (n + 63) // 64 * 64
            lay = _LOSS_LAYOUTS[tuple(shapes)] = (offs, sizes, [ops._contig_strides(tuple(s)) for s in shapes], tot)
        offs, sizes, strides, tot = lay
        flat = torch.zeros(tot, device=dev)
        bufs = [flat.as_strided(s, st, o) for o, s, st in zip(offs, shapes, strides)]
        L, Lp, grads = bufs[0], bufs[1], bufs[2:]
        gi = iter(grads)
        g_per = [(next(gi), [next(gi), next(gi), next(gi)]) for _ in per]
        # ---- pixel terms: in the first branch's loss launch (tn_train_losses)
        if spec.separate:
            (_, _, _, comp, _), (_, _, _, comp_t, _) = per
            pixel = (comp, comp_t, image, is_thermal, spec.thermal_mult, spec.tv_mult, spec.cross_mult, g_per[0][0], g_per[1][0])
        else:
            comp, d_comp = per[0][3], g_per[0][0]
            pixel = (comp[:, :3], comp[:, 3:], image, is_thermal, spec.thermal_mult, spec.tv_mult, spec.cross_mult, d_comp[:, :3], d_comp[:, 3:])
        # ---- proposal terms (metrics_dict["distortion"] is the sum over suffixes and enters once per suffix: x nsfx, :363-368)
        nsfx = len(per)
        for (sfx, sb, pg, comp, ws), (_, dws) in zip(per, g_per):
            w = [x[..., 0] for x in ws]
            ops.train_losses(sb[2], w[2], [(sb[i], w[i], dws[i] if pg else None) for i in range(2)], spec.distortion_mult * nsfx, spec.interlevel_mult,
                             dws[2], Lp, pixel=pixel)
            pixel = None
        # ---- density cross terms  a*|d2.detach - dens_t| + b*|d2 - dens_t.detach|  and  a*|dens.detach - d2t| + b*|dens - d2t.detach|
        if dens:
            a, b = spec.density_weights
            gd = [next(gi) for _ in range(4)]
            ops.l1_loss(dens[0], dens[1], b, a, L[10:11], gd[0], gd[1])
            ops.l1_loss(dens[2], dens[3], b, a, L[10:11], gd[2], gd[3])
        for k, (pose, (tp, rp, sc)) in enumerate(zip(poses, spec.camera_regs)):
            if k == 0:  # the loss lines are added up in the first regulariser's launch
                ops.losses_finish(Lp, L, pose.detach(), tp, rp, sc, L[11:12], next(gi))
            else:
                ops.camera_reg(pose.detach(), tp, rp, sc, L[11 + k:12 + k], next(gi))
        if not poses:
            ops.losses_finish(Lp, L)
        ctx.set_materialize_grads(False)  # unused terms arrive as None, not as zeros
        # the terms this iteration actually produced (a term that does not exist in this mode is never waited for in backward)
        ctx.live = [0, 1, 2, 3, 8, 9] + ([10] if dens else []) + [11 + k for k in range(len(poses))]
        ctx.n_inputs = 3 + len(ts)
        ctx.flat, ctx.layout, ctx.nper = flat, (offs[2:], shapes[2:], strides[2:]), [pg for _, _, pg, _, _ in per]
        ops.train_metrics(L, image.shape[0], spec.thermal_mult, [p.detach() for p in poses], L[16:24])
        outs = L.unbind(0)
        ctx.mark_non_differentiable(*(outs[i] for i in range(32) if i not in (0, 1, 2, 3, 8, 9, 10, 11, 12)))
        return outs

    @staticmethod
    def backward(ctx, *gs):
        g = _scalar_grads_equal([gs[i] for i in ctx.live], "TrainLosses")
        if g is None:
            return (None,) * ctx.n_inputs
        scaled = ctx.flat * g  # every gradient buffer at once (out of place: the loss values the caller holds live in the same allocation)
        out = [None, None, None]
        gi = iter([scaled.as_strided(s, st, o) for o, s, st in zip(*ctx.layout)])
        for pg in ctx.nper:
            out.append(next(gi))
            dws = [next(gi), next(gi), next(gi)]
            out += [dws[0].unsqueeze(-1) if pg else None, dws[1].unsqueeze(-1) if pg else None, dws[2].unsqueeze(-1)]
        out += list(gi)
        return tuple(out)
