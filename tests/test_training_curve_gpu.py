"""Multi-step training parity: the HIP path replays the oracle's 120-iteration training run (tests/golden/train_curve_shared.npz, produced by
oracle/make_train_curve.py with the reference Trainer's ordering) on identical rays, targets and per-step jitters.

What it pins beyond the single-step tests: Adam state across steps (per-group step counts; the proposal networks are only stepped on the
iterations where the sampler gave them gradients), the exponential LR schedules, the proposal update schedule, weight annealing.
Training is a chaotic map (see helpers.oracle_grad_sensitivity): the early steps must agree tightly, later ones to a few percent."""
import os

import numpy as np
import pytest
import torch

import make_train_curve as mtc
from nerfstudio_thermal_amd import synth
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.engine import RenderEngine
from test_hip_ops_gpu import pkg_cfg
import thermal_nerfacto_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_training_curve_matches_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_curve_shared.npz"))
    T, N = int(g["steps"]), int(g["num_rays"])
    ocfg = orc.OracleConfig(density_mode="shared", **mtc.TINY)
    cfg = pkg_cfg(ocfg)
    arena = ParamArena(cfg, ocfg.num_images, DEV)
    arena.load({k: torch.from_numpy(v) for k, v in synth.synth_params(orc.param_shapes(ocfg), seed=0, table_scale=0.1).items()})
    eng = RenderEngine(cfg, arena, ocfg.num_images, list(ocfg.is_thermal_cam))
    idx, o, d, img, is_th = mtc.inputs(N)
    o, d, cam, img, is_th = o.to(DEV).contiguous(), d.to(DEV).contiguous(), idx[:, 0].to(DEV).contiguous(), img.to(DEV), is_th.to(DEV)
    totals, updated = [], []
    for step in range(T):
        jit = [torch.from_numpy(j).to(DEV).reshape(-1) for j in synth.synth_jitters(N, seed=1000 + step)]
        was_updated = eng.steps_since_update > eng.update_schedule(eng.sampler_step) or eng.sampler_step < 10
        losses = eng.train_step(o, d, cam, img, is_th, step, jitters=jit)
        totals.append(float(sum(losses.values())))
        updated.append(int(was_updated))
    ref = g["curve/total"]
    totals = np.array(totals)
    assert np.array_equal(np.array(updated), g["updated"])  # same proposal-update schedule
    rel = np.abs(totals - ref) / np.abs(ref)
    # envelope: the oracle's own deviation when its jitters move by one fp32 ulp (stored by make_train_curve.py); training is chaotic
    # (Adam eps=1e-15 turns rounding-level gradient differences into lr-sized steps), so the HIP path is held to a small multiple of it
    env = np.abs(g["curve_perturbed/total"] - ref) / np.abs(ref)
    env_run = np.maximum.accumulate(env)
    assert rel[:10].max() <= 3e-3, rel[:10]
    assert bool((rel <= 4.0 * env_run + 2e-2).all()), (float(rel.max()), int(rel.argmax()), float(env_run[int(rel.argmax())]))
    assert 0.5 <= totals[-1] / ref[-1] <= 2.0 and totals[-1] < totals[0] / 500  # it trained as far as the oracle did (loss fell > 500x)
    out, _ = eng.get_outputs(o, d, cam, training=False)
    # The two trained models are different samples of a chaotic process (and the HIP run is not even reproducible run to run: float atomics),
    # so the renders are compared two ways: (1) both must fit the training targets equally well, (2) they must agree with each other about
    # as well as oracle-vs-perturbed-oracle does (10 dB slack: the spread of that PSNR over repeated HIP runs is ~6 dB).
    gt = img.cpu().double()
    th = is_th.cpu() > 0
    for key, mask, tgt in (("rgb", ~th, gt), ("rgb_thermal", th, gt[:, :1])):
        mine = out[key].cpu().double()
        theirs = torch.from_numpy(g[f"eval_{key}"]).double()
        fit_mine = float(((mine - tgt)[mask] ** 2).mean())
        fit_ref = float(((theirs - tgt)[mask] ** 2).mean())
        assert fit_mine <= 2.0 * fit_ref + 1e-6, (key, fit_mine, fit_ref)  # within 3 dB of the oracle's fit of the targets
        mse = float(((mine - theirs) ** 2).mean())
        mse_env = float(((torch.from_numpy(g[f"eval_{key}_perturbed"]).double() - theirs) ** 2).mean())
        psnr, psnr_env = -10 * np.log10(max(mse, 1e-30)), -10 * np.log10(max(mse_env, 1e-30))
        assert psnr >= min(psnr_env - 10.0, 40.0), (key, psnr, psnr_env)
