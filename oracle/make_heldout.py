"""Golden HELD-OUT quality from the oracle (CPU; test infrastructure): 300 training iterations at 1024 rays on the TRAIN split of the synthetic
RGB+T cube scene on disk (oracle/heldout_common.py), the reference Trainer's ordering as in oracle/make_train_curve.py, then the eval-mode render
(mean appearance embedding, no pose correction: models/thermal_nerfacto.py:403-489 at inference, pipelines/base_pipeline.py:377-440) of the
HELD-OUT cameras on a pixel grid, PSNR per spectrum.  Stored twice: the run itself and the same run with every jitter moved by one fp32 ulp --
the oracle's own spread.  tests/test_heldout_quality_gpu.py replays the schedule through the HIP path.

    python oracle/make_heldout.py        # writes tests/golden/heldout_shared.json   (~6 min on 8 cores)
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import heldout_common as hc  # noqa: E402
import thermal_nerfacto_oracle as orc  # noqa: E402
from make_train_curve import anneal_for, update_schedule  # noqa: E402
from nerfstudio_thermal_amd import synth  # noqa: E402


def heldout_psnr(params, cfg, va, va_images):
    t = lambda k: va.cameras[k]  # noqa: E731
    out = {"rgb": [], "thermal": []}
    for i in range(len(va_images)):
        idx = hc.eval_indices(va, i)
        o, d, _, _ = orc.generate_rays(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
        with torch.no_grad():
            ev = orc.get_outputs(params, cfg, o, d, torch.zeros(idx.shape[0], dtype=torch.int64), training=False)
        gt = va_images[i][idx[:, 1], idx[:, 2]]
        if va.metadata["is_thermal"][i]:
            out["thermal"].append(hc.psnr(ev["rgb_thermal"], gt[:, :1]))
        else:
            out["rgb"].append(hc.psnr(ev["rgb"], gt))
    return {k: float(np.mean(v)) for k, v in out.items()}


def run(data_dir, perturb):
    torch.set_num_threads(8)
    tr, tr_images, va, va_images = hc.splits(data_dir)
    is_th = [int(x) for x in tr.metadata["is_thermal"]]
    cfg = orc.OracleConfig(density_mode="shared", num_images=len(tr_images), is_thermal_cam=tuple(is_th), **hc.TINY)
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.synth_params(orc.param_shapes(cfg), seed=0, table_scale=0.1).items()}
    groups = orc.optimizer_groups(cfg)
    sched = {"proposal_networks": (1e-2, 1e-4, 200000), "fields": (1e-2, 1e-4, 200000), "camera_opt": (1e-3, 1e-4, 5000)}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
    th_pos = torch.tensor([float(x) for x in is_th])
    order = torch.arange(len(tr_images))
    t = lambda k: tr.cameras[k]  # noqa: E731
    sampler_step, since_update, totals = 0, 0, []
    for step in range(hc.T_STEPS):
        u = torch.from_numpy(hc.step_uniforms(step))
        idx, img, is_thermal = orc.sample_pixels(tr_images, th_pos, order, hc.N_RAYS, u, 2)
        o, d, _, _ = orc.generate_rays(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
        jit = [torch.from_numpy(j) for j in hc.step_jitters(step)]
        if perturb:
            jit = [torch.nextafter(j, torch.tensor(2.0)) for j in jit]
        updated = since_update > update_schedule(sampler_step) or sampler_step < 10
        out = orc.get_outputs(params, cfg, o, d, idx[:, 0].contiguous(), training=True, anneal=anneal_for(step), jitters=jit, prop_requires_grad=updated)
        if updated:
            since_update = 0
        losses = orc.loss_dict(params, cfg, out, img, is_thermal, training=True)
        total = sum(losses.values())
        total.backward()
        totals.append(float(total))
        with torch.no_grad():
            for gname, (keys, _) in groups.items():
                lr0, lr_final, max_steps = sched[gname]
                lr = orc.exp_decay_lr(step, lr0, lr_final, max_steps)
                for k in keys:
                    p = params[k]
                    if p.grad is None:
                        continue
                    orc.adam_step(p, p.grad, state[k][0], state[k][1], step + 1, lr)
                    p.grad = None
        sampler_step = step
        since_update += 1
        if step % 25 == 0:
            print(step, round(totals[-1], 5), flush=True)
    return totals, heldout_psnr(params, cfg, va, va_images)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        hc.write_scene(tmp)
        totals, ps = run(tmp, False)
        ptotals, pps = run(tmp, True)
    res = {"steps": hc.T_STEPS, "num_rays": hc.N_RAYS, "frames_per_spectrum": hc.FRAMES, "tables": hc.TINY,
           "heldout_psnr": ps, "heldout_psnr_perturbed": pps, "total_loss_first": totals[0], "total_loss_last": totals[-1],
           "total_loss_last_perturbed": ptotals[-1],
           "note": "oracle (CPU) trained on the train split of the synthetic cube scene on disk; PSNR of the HELD-OUT cameras (every 4th image) on a pixel grid; "
                   "_perturbed: the same run with every sampler jitter moved by one fp32 ulp"}
    with open(os.path.join(hc.ROOT, "tests", "golden", "heldout_shared.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
