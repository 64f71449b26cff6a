import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import make_train_curve as mtc
from nerfstudio_thermal_amd import synth
from nerfstudio_thermal_amd.arena import ParamArena
from nerfstudio_thermal_amd.engine import RenderEngine
from test_hip_ops_gpu import pkg_cfg
import thermal_nerfacto_oracle as orc
g = np.load(os.path.join(ROOT, "tests/golden/train_curve_shared.npz"))
T, N = int(g["steps"]), int(g["num_rays"])
ocfg = orc.OracleConfig(density_mode="shared", **mtc.TINY); cfg = pkg_cfg(ocfg)
arena = ParamArena(cfg, 8, "cuda")
arena.load({k: torch.from_numpy(v) for k, v in synth.synth_params(orc.param_shapes(ocfg), seed=0, table_scale=0.1).items()})
eng = RenderEngine(cfg, arena, 8, list(ocfg.is_thermal_cam))
idx, o, d, img, is_th = mtc.inputs(N)
o, d, cam, img, is_th = o.cuda().contiguous(), d.cuda().contiguous(), idx[:, 0].cuda().contiguous(), img.cuda(), is_th.cuda()
keys = ["rgb_loss", "thermal_loss", "interlevel_loss", "distortion_loss", "camera_opt_regularizer"]
for step in range(T):
    jit = [torch.from_numpy(j).cuda().reshape(-1) for j in synth.synth_jitters(N, seed=1000 + step)]
    losses = eng.train_step(o, d, cam, img, is_th, step, jitters=jit)
    tot = float(sum(losses.values()))
    print(step, int(g["updated"][step]), f"total {tot:.6g} ref {g['curve/total'][step]:.6g} rel {abs(tot-g['curve/total'][step])/g['curve/total'][step]:.2e} | " +
          " ".join(f"{k[:5]} {abs(float(losses[k])-g['curve/'+k][step])/max(abs(g['curve/'+k][step]),1e-12):.1e}" for k in keys))
