"""Golden TRAINING CURVE from the oracle (CPU; test infrastructure).

Runs T training iterations of the oracle with the reference Trainer's ordering (engine/trainer.py:224-290,455-499):
  BEFORE_TRAIN_ITERATION set_anneal(step) -> forward (proposal nets under no_grad unless `updated`) -> losses -> backward ->
  Adam per param group with the ExponentialDecay lr of that step -> AFTER_TRAIN_ITERATION step_cb(step)
on deterministic inputs (synthetic rays, smooth synthetic ground truth, per-step jitters from the integer-hash generator) and stores the
loss curve plus an eval render with the trained weights.  tests/test_training_curve_gpu.py replays the same schedule through the HIP path.

    python oracle/make_train_curve.py            # writes tests/golden/train_curve_shared.npz   (~2 min on 8 cores)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import nerfstudio_thermal_amd  # noqa: E402,F401
from nerfstudio_thermal_amd import synth  # noqa: E402
import thermal_nerfacto_oracle as orc  # noqa: E402

T_STEPS = 120
N_RAYS = 512
TINY = dict(log2_hashmap_size=12, prop_log2_hashmap_size=10)


def inputs(num_rays=N_RAYS):
    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, num_rays, seed=5)
    img, is_th = synth.synth_gt_smooth(idx, cams)
    t = lambda k: torch.from_numpy(cams[k])  # noqa: E731
    o, d, _, _ = orc.generate_rays(torch.from_numpy(idx), t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
    return torch.from_numpy(idx), o, d, torch.from_numpy(img), torch.from_numpy(is_th)


def update_schedule(step, warmup=5000, every=5):
    return float(np.clip(np.interp(step, [0, warmup], [0, every]), 1, every))


def anneal_for(step, n=1000, slope=10.0):
    frac = float(np.clip(step / n, 0, 1))
    return slope * frac / ((slope - 1) * frac + 1)


def run(mode, perturb):
    """perturb=True: the same run with every jitter moved by one fp32 ulp -- the oracle's OWN chaotic divergence, stored as the envelope
    against which the HIP path's deviation is judged (training with Adam eps=1e-15 amplifies rounding-level differences)."""
    torch.set_num_threads(8)
    cfg = orc.OracleConfig(density_mode=mode, **TINY)
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.synth_params(orc.param_shapes(cfg), seed=0, table_scale=0.1).items()}
    idx, o, d, img, is_th = inputs()
    cam = idx[:, 0]
    groups = orc.optimizer_groups(cfg)
    sched = {"proposal_networks": (1e-2, 1e-4, 200000), "fields": (1e-2, 1e-4, 200000), "camera_opt": (1e-3, 1e-4, 5000),
             "proposal_networks_thermal": (1e-2, 1e-4, 200000), "fields_thermal": (1e-2, 1e-4, 200000), "camera_opt_thermal": (1e-3, 1e-4, 5000)}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
    sampler_step, since_update = 0, 0
    curve = {}
    updated_flags = []
    for step in range(T_STEPS):
        anneal = anneal_for(step)
        updated = since_update > update_schedule(sampler_step) or sampler_step < 10
        jit = [torch.from_numpy(j) for j in synth.synth_jitters(N_RAYS, seed=1000 + step)]
        if perturb:
            jit = [torch.nextafter(j, torch.tensor(2.0)) for j in jit]
        out = orc.get_outputs(params, cfg, o, d, cam, training=True, anneal=anneal, jitters=jit, prop_requires_grad=updated)
        if updated:
            since_update = 0
        updated_flags.append(int(updated))
        losses = orc.loss_dict(params, cfg, out, img, is_th, training=True)
        total = sum(losses.values())
        total.backward()
        for k, v in losses.items():
            curve.setdefault(k, []).append(float(v))
        curve.setdefault("total", []).append(float(total))
        with torch.no_grad():
            for gname, (keys, lr0) in groups.items():
                lr0, lr_final, max_steps = sched[gname]
                lr = orc.exp_decay_lr(step, lr0, lr_final, max_steps)
                for k in keys:
                    p = params[k]
                    if p.grad is None:
                        # torch.optim skips parameters without a gradient; the flat-arena Adam of the HIP path sees an exact zero gradient there:
                        # identical unless the moments are non-zero (then Adam keeps moving the parameter).  Mirror torch.optim here.
                        continue
                    orc.adam_step(p, p.grad, state[k][0], state[k][1], step + 1, lr)
                    p.grad = None
        sampler_step = step
        since_update += 1
        if step % 20 == 0:
            print(step, {k: round(v[-1], 6) for k, v in curve.items()}, flush=True)
    with torch.no_grad():
        ev = orc.get_outputs(params, cfg, o, d, cam, training=False)
    return curve, updated_flags, ev


def main(mode="shared"):
    curve, updated_flags, ev = run(mode, False)
    pcurve, _, pev = run(mode, True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"train_curve_{mode}.npz"), steps=T_STEPS, num_rays=N_RAYS, updated=np.array(updated_flags),
                        eval_rgb=ev["rgb"].numpy(), eval_rgb_thermal=ev["rgb_thermal"].numpy(),
                        eval_rgb_perturbed=pev["rgb"].numpy(), eval_rgb_thermal_perturbed=pev["rgb_thermal"].numpy(),
                        **{f"curve/{k}": np.array(v) for k, v in curve.items()}, **{f"curve_perturbed/{k}": np.array(v) for k, v in pcurve.items()})
    rel = np.abs(np.array(pcurve["total"]) - np.array(curve["total"])) / np.array(curve["total"])
    print("final", {k: v[-1] for k, v in curve.items()})
    print("oracle-vs-perturbed-oracle relative deviation of the total loss: first 10 max %.2e, first 40 max %.2e, overall max %.2e" % (rel[:10].max(), rel[:40].max(), rel.max()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "shared")
