"""Which stage of the proposal sampler moves the HIP path's resampled bins off the reference's, bit-wise?  ("attribute the sampler's ulps")

The train-mode forward of tests/golden/model_shared_default256.npz (outputs of the reference itself: 256 rays, default tables, the golden's jitter,
anneal of step 500) is re-run stage by stage with the ORACLE on the CPU (checked here to reproduce the golden's bins bit for bit), and ONE stage at
a time is replaced by the HIP library's result:

    level-0 bins -> proposal density -> get_weights -> pow(w, anneal) -> PDF resampling (padding, pdf, cdf, searchsorted, lerp) -> level-1 bins

The table prints, per substitution, the share of level-1 euclidean bins that stay bit-identical to the reference's, within 4 ulps, and beyond 64.
The oracle is the CHECKER here (test infrastructure): nothing of it is on the product path.

    python scripts/sampler_ulps.py [out.md]          (needs the GPU)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch

import nerfstudio_thermal_amd  # noqa: F401
import thermal_nerfacto_oracle as orc
from nerfstudio_thermal_amd import ops, synth

import bench


def ulps(a: torch.Tensor, b: torch.Tensor):
    ia = a.detach().cpu().contiguous().view(torch.int32).to(torch.int64)
    ib = b.detach().cpu().contiguous().view(torch.int32).to(torch.int64)
    d = (ia - ib).abs()
    return {"identical": float((d == 0).double().mean()), "within_4ulp": float((d <= 4).double().mean()), "beyond_64ulp": float((d > 64).double().mean()),
            "max_ulp": int(d.max())}


def main():
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_shared_default256.npz"))
    cfg, arena, eng = bench.build_engine(dev)
    ocfg = orc.OracleConfig(density_mode="shared")
    params = {k: torch.from_numpy(v) for k, v in synth.synth_params(orc.param_shapes(ocfg), seed=0).items()}
    o, d = torch.from_numpy(g["rays/origins"]), torch.from_numpy(g["rays/directions"])
    cam = torch.from_numpy(g["rays/camera_indices"].astype(np.int64))[:, 0].contiguous()
    n = int(g["num_rays"])
    jit = [torch.from_numpy(j).reshape(-1, 1) for j in synth.synth_jitters(n)]
    eng.set_anneal_for_step(500)
    anneal = float(eng.anneal)
    assert abs(anneal - float(g["train/anneal"])) < 1e-12
    nears, fars = torch.full((n, 1), ocfg.near_plane), torch.full((n, 1), ocfg.far_plane)
    # pose correction as the training forward applies it (the oracle's own: the HIP result is bit-identical per tests/test_hip_ops_gpu.py)
    with torch.no_grad():
        po, pd = orc.apply_pose_adjustment(params["camera_optimizer.pose_adjustment"], torch.tensor(ocfg.is_thermal_cam, dtype=torch.bool), cam, o, d)
    S0, S1, S2 = eng.counts
    G = lambda t: t.to(dev).contiguous()  # noqa: E731
    with torch.no_grad():
        s0 = orc.spaced_bins(n, S0, jit[0]).contiguous()
        e0 = orc.s_to_euclidean(s0, nears, fars)
        smp0 = orc.Samples(s_bins=s0, e_bins=e0)
        ref_e0 = torch.from_numpy(g["train/ebins_0"])
        rows = [("level-0 bins (oracle vs reference golden)", ulps(e0, ref_e0))]
        dens_o = orc.prop_density(params, "proposal_networks", 0, ocfg, smp0.positions(po, pd))
        w_o = orc.get_weights(smp0.deltas, dens_o)
        pw_o = torch.pow(w_o, anneal)
        s1_o = orc.pdf_resample(s0, pw_o, S1, jit[1])
        e1_o = orc.s_to_euclidean(s1_o, nears, fars)
        ref_e1 = torch.from_numpy(g["train/ebins_1"])
        rows.append(("level-1 bins, every stage by the oracle (vs reference golden)", ulps(e1_o, ref_e1)))

        def finish(dens=None, w=None, pw=None, hip_pdf=False, hip_pow=False):
            """the rest of the chain by the oracle behind what was substituted; hip_pdf: the PDF stage by the library"""
            if w is None:
                w = orc.get_weights(smp0.deltas, dens if dens is not None else dens_o)
            if pw is None:
                pw = torch.pow(G(w), anneal).cpu() if hip_pow else torch.pow(w, anneal)
            if hip_pdf:
                _, e1 = ops.pdf_resample(G(s0), G(pw[..., 0]), S1, 1.0, G(nears.reshape(-1)), G(fars.reshape(-1)), G(jit[1].reshape(-1)))
                return e1.cpu()
            return orc.s_to_euclidean(orc.pdf_resample(s0, pw, S1, jit[1]), nears, fars)

        dens_h = ops.prop_density_fwd(eng.props[0], G(po), G(pd), G(e0)).cpu()[..., None]
        rows.append(("proposal density by HIP (k_prop_fwd), rest oracle", ulps(finish(dens=dens_h), ref_e1)))
        rows.append(("   ... the densities themselves (HIP vs oracle)", ulps(dens_h, dens_o)))
        w_h, _ = ops.weights_fwd(G(e0), G(dens_o[..., 0]))
        w_h = w_h.cpu()[..., None]
        rows.append(("get_weights by HIP (on the oracle's densities), rest oracle", ulps(finish(w=w_h), ref_e1)))
        rows.append(("   ... the weights themselves (HIP vs oracle)", ulps(w_h, w_o)))
        pw_h = torch.pow(G(w_o), anneal).cpu()  # ocml powf on the device: what the kernel's powf evaluates (checked two rows below)
        rows.append(("pow(w, anneal) on the device (ocml powf), rest oracle", ulps(finish(pw=pw_h), ref_e1)))
        rows.append(("   ... the powers themselves (device vs host)", ulps(pw_h, pw_o)))
        _, e1_a = ops.pdf_resample(G(s0), G(w_o[..., 0]), S1, anneal, G(nears.reshape(-1)), G(fars.reshape(-1)), G(jit[1].reshape(-1)))
        _, e1_b = ops.pdf_resample(G(s0), G(pw_h[..., 0]), S1, 1.0, G(nears.reshape(-1)), G(fars.reshape(-1)), G(jit[1].reshape(-1)))
        rows.append(("   (check: tn_pdf_resample(w, anneal) == tn_pdf_resample(device pow(w), 1))", ulps(e1_a, e1_b)))
        rows.append(("PDF stage by HIP (padding, pdf, cdf, search, lerp, s->euclid) on the oracle's pow(w)", ulps(finish(pw=pw_o, hip_pdf=True), ref_e1)))
        rows.append(("pow + PDF stage by HIP (tn_pdf_resample on the oracle's weights)", ulps(e1_a.cpu(), ref_e1)))
        # inside the PDF stage, on the CPU: the oracle's algorithm with its cumsum replaced by what the kernel computes (double accumulate, one rounding
        # per element) -- torch's CPU cumsum of float32 is NOT that everywhere (it adds float partial sums of vector chunks)
        w = pw_o[..., 0] + 0.01
        w_sum = torch.sum(w, dim=-1, keepdim=True)
        padding = torch.relu(1e-5 - w_sum)
        w = w + padding / w.shape[-1]
        w_sum = w_sum + padding
        pdf = w / w_sum
        cdf_t = torch.cumsum(pdf, dim=-1)
        cdf_d = torch.cumsum(pdf.double(), dim=-1).float()
        rows.append(("   torch.cumsum(pdf) float32 vs double-accumulated-then-rounded (host only)", ulps(cdf_t, cdf_d)))
        wsum_d = w_o[..., 0].double().add(0.01).sum(-1, keepdim=True).float()  # (the kernel sums (w + 0.01) in double)
        rows.append(("   sum(w + 0.01): torch float32 sum vs double-accumulated (host only)", ulps(torch.sum(pw_o[..., 0] + 0.01, dim=-1, keepdim=True), (pw_o[..., 0] + 0.01).double().sum(-1, keepdim=True).float())))
        # whole chain by HIP (what the product runs), for the record
        _, br = eng.get_outputs(G(o), G(d), G(cam), True, [G(j.reshape(-1)) for j in jit], None)
        rows.append(("whole chain by HIP: level 1", ulps(br[""].levels[1].e_bins, ref_e1)))
        rows.append(("whole chain by HIP: level 2", ulps(br[""].levels[2].e_bins, torch.from_numpy(g["train/ebins_2"]))))
    lines = ["# Sampler ulps: which stage moves the resampled bins (scripts/sampler_ulps.py, round 6)", "",
             "Level-1 euclidean bins of `tests/golden/model_shared_default256.npz` (256 rays x 97 bins, train forward, anneal of step 500) against the",
             "REFERENCE's; one stage of the oracle's chain replaced by the HIP library's result per row.", "",
             "| what ran on HIP | bit-identical | within 4 ulp | beyond 64 ulp | max ulp |", "|---|---|---|---|---|"]
    for name, r in rows:
        lines.append(f"| {name} | {r['identical']:.4f} | {r['within_4ulp']:.4f} | {r['beyond_64ulp']:.5f} | {r['max_ulp']} |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
