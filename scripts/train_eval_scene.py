"""N3 end to end on an RGB+T dataset ON DISK: write the synthetic cube scene in the reference's transforms.json format (or take --data), train
thermal-nerfacto on it with the fused step, render every image of the val split in full and report PSNR / SSIM per spectrum.

    python scripts/train_eval_scene.py [--data DIR] [--steps 3000] [--frames 12]
"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nerfstudio_thermal_amd  # noqa: F401
from nerfstudio_thermal_amd import ops, synth
from nerfstudio_thermal_amd.dataparser import write_rgbt_dataset
from nerfstudio_thermal_amd.pipeline import ThermalPipeline


def write_cube_scene(out_dir, frames_per_spectrum, device):
    """cameras on a ring looking at the cube; the thermal camera of a pair sits 5 cm beside its RGB camera (a rig), images rendered analytically"""
    cams = synth.synth_cameras(frames_per_spectrum, frames_per_spectrum)
    n = frames_per_spectrum
    cams["c2w"][n:] = cams["c2w"][:n]
    cams["c2w"][n:, :, 3] += 0.05 * cams["c2w"][:n, :, 0]
    t = lambda k: torch.from_numpy(cams[k]).to(device)  # noqa: E731
    images = []
    for c in range(2 * n):
        H, W = int(cams["height"][c]), int(cams["width"][c])
        yy, xx = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
        idx = torch.stack([torch.full((H * W,), c, device=device), yy.reshape(-1), xx.reshape(-1)], 1).contiguous()
        o, d, _, _ = ops.raygen(idx, t("c2w"), t("fx"), t("fy"), t("cx"), t("cy"), t("distortion"))
        images.append(synth.cube_scene_images(o.cpu().numpy(), d.cpu().numpy(), bool(cams["is_thermal"][c])).reshape(H, W, 3))
    return write_rgbt_dataset(out_dir, cams, images)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--frames", type=int, default=12)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    tmp = None
    data = args.data
    if data is None:
        tmp = tempfile.TemporaryDirectory()
        data = tmp.name
        write_cube_scene(data, args.frames, dev)
    pipe = ThermalPipeline(data, device=dev)
    curve = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        k = min(500, args.steps - done)
        losses = pipe.train(k)
        done += k
        torch.cuda.synchronize()
        curve.append({"step": done, "seconds": time.perf_counter() - t0, **losses})
    train_s = time.perf_counter() - t0
    pipe.get_average_eval_image_metrics()  # (first pass: allocations, workspaces -- the second one is timed, as TEST_RAYS_PER_SEC is a steady-state figure)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    metrics = pipe.get_average_eval_image_metrics()
    torch.cuda.synchronize()
    eval_rays = int(sum(int(w) * int(h) for w, h in zip(pipe.eval_outputs.cameras["width"], pipe.eval_outputs.cameras["height"])))
    print(json.dumps({"dataset": "synthetic cube scene on disk (transforms.json, RGB frames first, per-frame intrinsics, is_thermal)" if tmp else data,
                      "train_images": len(pipe.train_outputs.image_filenames), "eval_images": len(pipe.eval_outputs.image_filenames), "steps": args.steps,
                      "train_seconds": train_s, "train_rays_per_s": args.steps * 4096 / train_s, "eval_seconds": time.perf_counter() - t1, "eval_rays": eval_rays,
                      "eval_rays_per_s_incl_image_loading_and_metrics": eval_rays / (time.perf_counter() - t1),
                      "eval_metrics": metrics, "curve": curve}))


if __name__ == "__main__":
    main()
