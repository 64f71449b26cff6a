"""CPU analysis: how many atomic requests does the table scatter issue under different merging strategies? (guides kernel design)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import nerfstudio_thermal_amd
from nerfstudio_thermal_amd import synth
import thermal_nerfacto_oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = orc.OracleConfig(density_mode="shared")
params = {k: torch.from_numpy(v) for k, v in synth.synth_params(orc.param_shapes(cfg), seed=0).items()}
cams = synth.synth_cameras()
idx = torch.from_numpy(synth.synth_ray_indices(cams, N, seed=42))
tc = {k: torch.from_numpy(cams[k]) for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
o, d, _, _ = orc.generate_rays(idx, tc["c2w"], tc["fx"], tc["fy"], tc["cx"], tc["cy"], tc["distortion"])
with torch.no_grad():
    out = orc.get_outputs(params, cfg, o, d, idx[:, 0], training=True, jitters=[torch.rand(N, 1) for _ in range(3)])
for lvl, (name, L, minr, maxr) in enumerate([("prop0", 5, 16, 128), ("prop1", 5, 16, 256), ("main", 16, 16, 2048)]):
    smp = out["samples_list"][lvl]
    pos = smp.positions(o, d)
    p, sel = orc.unit_cube_positions(pos)
    res = orc.level_resolutions(L, minr, maxr)
    S = p.shape[1]
    tot_naive = tot_run = tot_patch = tot_wave_sorted = 0
    per_level = []
    for l in range(L):
        f = torch.floor(p * res[l]).to(torch.int64)           # [N,S,3]
        key = (f[..., 0] * 4096 + f[..., 1]) * 4096 + f[..., 2]   # cell id
        naive = N * S
        runs = 1 + (key[:, 1:] != key[:, :-1]).sum(1)         # runs along each ray
        run_total = int(runs.sum())
        # merging within a 2x2 patch (4 rays): distinct cells among 4*S samples
        kp = key.reshape(N // 4, 4 * S)
        patch_total = sum(len(np.unique(row)) for row in kp.numpy())
        # merging within a 64-lane wave of consecutive (ray-major) samples, any order (sorted)
        kw = key.reshape(-1)
        pad = (-len(kw)) % 64
        kw = torch.cat([kw, kw[-1:].expand(pad)]).reshape(-1, 64)
        wave_total = sum(len(np.unique(row)) for row in kw.numpy())
        per_level.append((float(res[l]), naive / run_total, naive / patch_total, naive / wave_total))
        tot_naive += naive; tot_run += run_total; tot_patch += patch_total; tot_wave_sorted += wave_total
    print(name, "S", S, "overall reduction: run-merge %.2fx, patch-unique %.2fx, wave-unique %.2fx" % (tot_naive / tot_run, tot_naive / tot_patch, tot_naive / tot_wave_sorted))
    for r in per_level:
        print("   res %6.0f  run %.2fx  patch %.2fx  wave-unique %.2fx" % r)

print("\n--- exact simulation of k_grid_scatter's merging (patch order, 16 samples per wave, run tails) ---")
for lvl, (name, L, minr, maxr) in enumerate([("prop0", 5, 16, 128), ("prop1", 5, 16, 256), ("main", 16, 16, 2048)]):
    smp = out["samples_list"][lvl]
    pos = smp.positions(o, d)
    p, sel = orc.unit_cube_positions(pos)
    res = orc.level_resolutions(L, minr, maxr)
    S = p.shape[1]
    # patch order: groups of 4 rays, sample-major
    pp = p.reshape(N // 4, 4, S, 3).permute(0, 2, 1, 3).reshape(-1, 3)
    tot_req = 0
    for l in range(L):
        f = torch.floor(pp * res[l]).to(torch.int64)
        key = (f[:, 0] * 4096 + f[:, 1]) * 4096 + f[:, 2]
        kw = key.reshape(-1, 16)
        tails = 1 + (kw[:, 1:] != kw[:, :-1]).sum(1)
        ntails = int(tails.sum())
        cross = float(((f[:, 0] % 8) == 7).float().mean())
        req = ntails * 4 * (1 + cross)
        tot_req += req
        print(f"   {name} res {float(res[l]):6.0f}: samples/tail {len(key)/ntails:6.2f}  requests/sample {req/len(key):.2f}")
    print(f"{name}: total requests for N={N}: {tot_req/1e6:.2f} M -> for 4096 rays {tot_req*4096/N/1e6:.2f} M = {tot_req*4096/N/21e9*1e6:.0f} us at 21 G req/s")

print("\n--- window-size study (patch order, run tails within a window of W samples) ---")
for lvl, (name, L, minr, maxr) in enumerate([("prop0", 5, 16, 128), ("prop1", 5, 16, 256), ("main", 16, 16, 2048)]):
    smp = out["samples_list"][lvl]
    p, sel = orc.unit_cube_positions(smp.positions(o, d))
    res = orc.level_resolutions(L, minr, maxr)
    S = p.shape[1]
    pp = p.reshape(N // 4, 4, S, 3).permute(0, 2, 1, 3).reshape(-1, 3)
    for W in (16, 64, 256, 4 * S):
        tot = 0
        for l in range(L):
            f = torch.floor(pp * res[l]).to(torch.int64)
            key = (f[:, 0] * 4096 + f[:, 1]) * 4096 + f[:, 2]
            kw = key.reshape(-1, W)
            ntails = int((1 + (kw[:, 1:] != kw[:, :-1]).sum(1)).sum())
            cross = float(((f[:, 0] % 8) == 7).float().mean())
            tot += ntails * 4 * (1 + cross)
        print(f"{name}: window {W:5d}: {tot*4096/N/1e6:6.2f} M requests at 4096 rays -> {tot*4096/N/21e9*1e6:5.0f} us")
