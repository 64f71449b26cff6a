#!/usr/bin/env python3
"""Per-message cost of an RCCL all-reduce as this process sees it (HIP events around back-to-back async all-reduces of one size).
On a 1-rank group this is RCCL's launch + device copy path -- the floor every exchanged range pays regardless of the ring; on N ranks
(torchrun) it is the real ring time.    python scripts/rccl_latency.py            (1 rank)   /   torchrun --nproc-per-node N scripts/rccl_latency.py"""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib

par = importlib.import_module("nerfstudio-thermal_amd.parallel")


def main():
    if "RANK" not in os.environ:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(par.free_port()))
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    out = {}
    for mb in (0.004, 0.1, 1, 4, 8, 12, 16, 24, 64):
        n = int(mb * 1e6 / 4)
        x = torch.ones(n, device="cuda")
        for _ in range(5):
            dist.all_reduce(x, op=dist.ReduceOp.AVG)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            dist.all_reduce(x, op=dist.ReduceOp.AVG)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out[f"{mb}MB"] = {"us": round(us, 1), "algbw_GBps": round(n * 4 / us / 1e3, 1)}
    if rank == 0:
        print(json.dumps({"world": world, "all_reduce": out}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
