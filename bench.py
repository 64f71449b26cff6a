#!/usr/bin/env python3
"""bench.py -- train rays/s of thermal-nerfacto (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one Trainer.train_iteration of the reference (engine/trainer.py:455-499) on a synthetic batch: ray generation for 4096 rays per
GPU (configs[1]: density_mode=shared, 256/96 proposal + 48 field samples, 2^19/2^17 tables), forward, every loss, backward, gradient
all-reduce (N>1) and Adam over all parameter groups.  Inputs (cameras, ray indices, ground truth) are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

RAYS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_engine(device, mode="shared", seed=0):
    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import synth
    from nerfstudio_thermal_amd.arena import ParamArena
    from nerfstudio_thermal_amd.config import ThermalNerfactoModelConfig
    from nerfstudio_thermal_amd.engine import RenderEngine

    cfg = ThermalNerfactoModelConfig(density_mode=mode)
    arena = ParamArena(cfg, 8, device)
    shapes = {n: s for n, (_, s) in arena.layout.items()}
    arena.load({k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, seed=seed).items()})
    eng = RenderEngine(cfg, arena, 8, [0, 0, 0, 0, 1, 1, 1, 1])
    return cfg, arena, eng


def make_batch(device, num_rays, seed):
    """Pre-staged pixel batch (kernel_roofline and the parity-style fixed batch)."""
    from nerfstudio_thermal_amd import synth

    cams = synth.synth_cameras()
    idx = synth.synth_ray_indices(cams, num_rays, seed=seed)
    img, is_th = synth.synth_gt(idx, cams, seed=seed)
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    cam_t = {k: t(cams[k]) for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    return cam_t, t(idx), t(img), t(is_th)


def make_image_cache(device):
    """The 8 synthetic training images (4 RGB 640x480, 4 thermal 160x120) resident in HBM, in dataset order."""
    from nerfstudio_thermal_amd import ops, synth

    cams = synth.synth_cameras()
    imgs = [torch.from_numpy(im) for im in synth.synth_images(cams)]
    n = len(imgs)
    return ops.ImageCache.build(imgs, torch.from_numpy(cams["is_thermal"].astype(np.float32)), torch.arange(n), device)


def one_step(eng, cam_t, cache, num_rays, step, hook):
    # datamanager.next_train (data/datamanagers/base_datamanager.py:538-547): this step's 2x2 pixel patches over the jagged image list, their
    # ground truth and the rays, all on the device (nerfstudio_thermal_amd/data.py)
    dm = getattr(eng, "_bench_dm", None)
    if dm is None or dm.num_rays != num_rays or dm.cache is not cache:
        from nerfstudio_thermal_amd.data import DeviceDataManager

        dm = eng._bench_dm = DeviceDataManager(cache, cam_t, num_rays, 2)
    o, d, cam, img, is_th = dm.next_train(step)
    return eng.train_step(o, d, cam, img, is_th, step, grad_hook=hook)


def time_ms(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# HBM-side traffic per launch from rocprofv3 PMC passes (profiles/r01_pmc_summary.md, scripts/pmc_passes.sh; N = 4096, shared mode): read bytes =
# TCC_EA0_RDREQ x 64 B (the calibrated form for these 8-byte gathers; FETCH_SIZE under-reports wide streams 2x on gfx950), write bytes =
# WRITE_SIZE x 1024.  A scatter entry point = zero-fill of the replica scratch + k_grid_scatter + k_replica_reduce: all three are counted.
PMC_TRAFFIC_BYTES = {
    "k_grid_scatter(main grid)": (274.1e6 + 352.8e6) + (22.5e6 + 2.3e6) + 16.2e6,
    "k_grid_scatter(prop0 grid)": (47.1e6 + 193.2e6) + (13.8e6 + 2.7e6) + 24.0e6,
    "k_grid_scatter(prop1 grid)": (26.3e6 + 118.6e6) + (10.9e6 + 2.9e6) + 17.8e6,
    "k_field_encode": 346.4e6 + 26.4e6,
    "k_prop_fwd(level0)": 12.1e6 + 4.1e6,
    "k_prop_fwd(level1)": 12.4e6 + 1.5e6,
}
ATOMIC_REQ_PEAK = 21.0e9  # 64-B atomic requests/s, measured by scripts/microbench/atomic_shapes.hip on MI355X
PMC_ATOMIC_REQUESTS = {"k_grid_scatter(main grid)": 10.0e6, "k_grid_scatter(prop0 grid)": 5.45e6, "k_grid_scatter(prop1 grid)": 3.37e6}


def kernel_roofline(eng, cam_t, idx):
    """Live HIP-event timing (torch.cuda.Event on torch's current stream = the stream every kernel of this library is launched on) of the
    hash-grid gather / scatter kernels, each launched alone as a single-kernel C-ABI call.  Algorithmic bytes: gather =
    points x levels x 8 corners x 8 B; scatter-add = read-modify-write = 2 x that."""
    from nerfstudio_thermal_amd import ops

    N = idx.shape[0]
    o, d, _, _ = ops.raygen(idx, cam_t["c2w"], cam_t["fx"], cam_t["fy"], cam_t["cx"], cam_t["cy"], cam_t["distortion"])
    cam = idx[:, 0].contiguous()
    out, br = eng.get_outputs(o, d, cam, True)
    b = br[""]
    lv = b.levels
    rows = []
    for i in range(2):
        S = lv[i].S
        ms = time_ms(lambda i=i: ops.prop_density_fwd(eng.props[i], b.origins, b.directions, lv[i].e_bins))
        rows.append((f"k_prop_fwd(level{i})", ms, N * S * 5 * 8 * 8))
    d_o, d_d = torch.zeros_like(o), torch.zeros_like(d)
    grids = [("prop0 grid", eng.props[0], lv[0]), ("prop1 grid", eng.props[1], lv[1]), ("main grid", eng.field, lv[2])]
    for name, net, L in grids:
        ld = 16 if net.num_levels == 5 else 32
        g_enc = torch.randn((N * L.S, ld), device=o.device) * 1e-3
        ms = time_ms(lambda: ops.hash_scatter(net.table, net.grads["table"], net.num_levels, net.log2_hashmap_size, net.res, b.origins, b.directions,
                                              L.e_bins, g_enc, d_o, d_d))
        rows.append((f"k_grid_scatter({name})", ms, 2 * N * L.S * net.num_levels * 8 * 8))
    eng.arena.zero_grad()
    return rows


def cpu_baseline(num_rays, steps, threads):
    """Oracle (pure-PyTorch restatement of the reference torch path, pinned to reference goldens) timed on the host cores:
    forward + losses + backward + Adam, same workload definition, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import thermal_nerfacto_oracle as orc
    from nerfstudio_thermal_amd import synth

    torch.set_num_threads(threads)
    cfg = orc.OracleConfig(density_mode="shared")
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.synth_params(orc.param_shapes(cfg), seed=0).items()}
    cams = synth.synth_cameras()
    idx = torch.from_numpy(synth.synth_ray_indices(cams, num_rays, seed=42))
    img, is_th = (torch.from_numpy(a) for a in synth.synth_gt(idx.numpy(), cams, seed=42))
    tc = {k: torch.from_numpy(cams[k]) for k in ("c2w", "fx", "fy", "cx", "cy", "distortion")}
    groups = orc.optimizer_groups(cfg)
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
    times = []
    for step in range(steps + 1):
        t0 = time.perf_counter()
        o, d, _, _ = orc.generate_rays(idx, tc["c2w"], tc["fx"], tc["fy"], tc["cx"], tc["cy"], tc["distortion"])
        jit = [torch.rand(num_rays, 1) for _ in range(3)]
        out = orc.get_outputs(params, cfg, o, d, idx[:, 0], training=True, anneal=1.0, jitters=jit)
        losses = orc.loss_dict(params, cfg, out, img, is_th, training=True)
        sum(losses.values()).backward()
        with torch.no_grad():
            for _, (keys, lr) in groups.items():
                for k in keys:
                    p = params[k]
                    if p.grad is not None:
                        orc.adam_step(p, p.grad, state[k][0], state[k][1], step + 1, lr)
                        p.grad = None
        if step > 0:
            times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {"value": num_rays / t, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{steps} steps (after 1 warm-up) of the same {num_rays}-ray shared-density train step (fwd+losses+bwd+Adam), median {t:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="shared", choices=["shared", "separate"], help="density_mode (default = BASELINE configs[1]; separate = configs[2])")
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU per step (configs[2] uses 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=RAYS_PER_GPU)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads for the baseline (0 = min(host cores, 16): more threads make the"
                    " many small ATen ops of this path slower; measured on the 256-thread GPU-box host at 1024 rays: 8 -> 2427, 16 -> 2286, 32 -> 1906, 64 -> 1025,"
                    " 256 -> 24 rays/s)")
    ap.add_argument("--ops", action="store_true", help="print the per-kernel timing table to stderr")
    ap.add_argument("--dp-chunks", type=int, default=-1, help="N>1: level ranges of the main table exchanged separately (-1 = the default "
                    "schedule 2/4/4/3/2/1 levels, n = n equal ranges, 0 = one all-reduce after the backward)")
    ap.add_argument("--force-dp", action="store_true", help="diagnostic: run the N>1 schedule (phased backward + overlapped RCCL all-reduce) on a "
                    "1-rank process group, to see what the schedule itself costs")
    args = ap.parse_args()

    import nerfstudio_thermal_amd  # noqa: F401
    from nerfstudio_thermal_amd import _lib
    from nerfstudio_thermal_amd.parallel import GradAllReducer, OverlappedGradReducer, broadcast_params, init_distributed, rank_seed

    _lib.load()  # fail loudly if the HIP library is missing
    rank, local, world = init_distributed()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    cfg, arena, eng = build_engine(device, mode=args.mode)
    rays = args.rays
    broadcast_params(arena)
    torch.manual_seed(rank_seed(42, rank))  # every rank draws its own pixels (scripts/train.py:97)
    cam_t, idx, img, is_th = make_batch(device, rays, seed=rank_seed(42, rank))
    cache = make_image_cache(device)
    # N > 1: the gradient all-reduce (RCCL) is issued per level range of the main table while the backward is still running
    make_hook = lambda w: (OverlappedGradReducer(w) if args.dp_chunks < 0 else OverlappedGradReducer(w, level_chunks=args.dp_chunks)  # noqa: E731
                           if args.dp_chunks > 0 else GradAllReducer(w))
    hook = make_hook(world) if world > 1 else None
    if args.force_dp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
        hook = make_hook(1)
        if args.dp_chunks == 0:
            hook.world = 2  # GradAllReducer returns early at world 1: make it issue the collective (the 1/2 scale does not matter here)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    step = 0
    for _ in range(args.warmup):
        one_step(eng, cam_t, cache, rays, step, hook)
        step += 1
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = one_step(eng, cam_t, cache, rays, step, hook)
        step += 1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = {k: float(v) for k, v in losses.items()}
    assert all(np.isfinite(v) for v in final_loss.values()), final_loss

    if rank == 0:
        rows = kernel_roofline(eng, cam_t, idx)
        if args.ops:
            for name, ms, nbytes in rows:
                print(f"{name:60s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s algorithmic", file=sys.stderr)
        name, ms, nbytes = max(rows, key=lambda r: r[1])
        achieved = nbytes / (ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": PMC_TRAFFIC_BYTES.get(name) if (rays == 4096 and args.mode == "shared") else None, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms,
                    "all_kernels": {n: {"ms": m, "GB/s": bts / (m * 1e-3) / 1e9} for n, m, bts in rows}}
        if name in PMC_ATOMIC_REQUESTS:
            # what actually bounds the scatter: 64-byte atomic requests at the memory side (microbenchmarked peak 21 G requests/s)
            rate = PMC_ATOMIC_REQUESTS[name] / (ms * 1e-3)
            roofline["atomic_requests"] = {"per_launch": PMC_ATOMIC_REQUESTS[name], "achieved_per_s": rate, "peak_per_s": ATOMIC_REQ_PEAK, "frac": rate / ATOMIC_REQ_PEAK}
        result = {
            "metric": "train rays/sec (4096-ray batch, 96 samples/ray)",
            "value": world * rays * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"thermal-nerfacto density_mode={args.mode} train step (pixel sampling+raygen+fwd+losses+bwd+allreduce+Adam), {rays} rays/GPU, "
                                   "256/96 proposal + 48 field samples, hash 16x2^19x2 + 2x(5x2^17x2), 8 cameras (4 RGB + 4 thermal)",
                       "rays_per_gpu": rays, "parallelism": f"dp{world}", "final_loss": final_loss},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "shared":
            torch.cuda.synchronize()
            cores = args.cpu_threads or min(os.cpu_count() or 1, 16)
            result["cpu_baseline"] = cpu_baseline(args.cpu_rays, args.cpu_steps, cores)
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
