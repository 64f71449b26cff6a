"""The product path's data-parallel schedule guard (trainer.FusedTrainerMixin on N > 1: parallel.InRunScheduleGuard) on CPU: two gloo ranks, the
fused step itself replaced by recorded times (it needs the GPU) -- what is tested is the decision protocol `ns-train thermal-nerfacto-hip` runs:
six iterations per schedule, MAX over the ranks, one decision for everybody, logged once; the reducers are the real ones
(pipelines/base_pipeline.py:281-283 is the reference's counterpart: a DistributedDataParallel wrap, which has no schedule to pick)."""
import os
import socket
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nerfstudio_thermal_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, slow_rank, q):
    from nerfstudio_thermal_amd import trainer as T
    from nerfstudio_thermal_amd.parallel import GradAllReducer, OverlappedGradReducer

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    logged = []
    T._log_schedule, keep = (lambda dec, step: logged.append((dec["schedule"], step))), T._log_schedule
    tr = types.SimpleNamespace()
    guard = T._dp_guard(tr)
    assert guard is T._dp_guard(tr) and guard.measuring  # one guard per Trainer
    assert isinstance(guard.hooks["overlapped"], OverlappedGradReducer) and isinstance(guard.hooks["simple"], GradAllReducer)
    hooks = []
    for step in range(14):
        hooks.append(type(guard.hook).__name__)
        if guard.measuring:
            # this rank's wall time of the iteration: the overlapped schedule stalls on `slow_rank` only (the hardware-queue cliff hit one process)
            over = 5.0 if rank == slow_rank else 1.0
            ms = (over if guard.leg == "overlapped" else 1.1) + (3.0 if len(guard.times[guard.leg]) < 2 else 0.0)  # (first two of a leg: warm-up, skipped)
            guard.record(ms, step)
    keep_log = list(logged)
    T._log_schedule = keep
    # the real log function: no nerfstudio writer here -> the console line, no exception
    T._log_schedule(guard.decision, 11)
    os.environ["TN_DP_SCHEDULE"] = "simple"  # pinned: no measurement
    pinned = T._dp_guard(types.SimpleNamespace())
    assert not pinned.measuring and type(pinned.hook).__name__ == "GradAllReducer" and pinned.decision["pinned_by"] == "TN_DP_SCHEDULE"
    del os.environ["TN_DP_SCHEDULE"]
    q.put((rank, guard.decision["schedule"], guard.decision["overlapped_ms"], guard.decision["simple_ms"], hooks, keep_log))
    dist.barrier()
    dist.destroy_process_group()


def _run(slow_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, slow_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_one_slow_rank_switches_every_rank_to_the_simple_schedule():
    res = _run(slow_rank=1)
    for rank, pick, over, simple, hooks, logged in res:
        assert pick == "simple" and over == 5.0 and simple == 1.1, res  # MAX over the ranks: rank 0 measured 1.0 itself
        assert hooks == ["OverlappedGradReducer"] * 6 + ["GradAllReducer"] * 8, hooks
        assert logged == [("simple", 11)], logged  # decided by the 12th iteration, logged once


def test_a_healthy_overlapped_schedule_is_kept():
    res = _run(slow_rank=-1)
    for rank, pick, over, simple, hooks, logged in res:
        assert pick == "overlapped" and over == 1.0, res
        assert hooks == ["OverlappedGradReducer"] * 6 + ["GradAllReducer"] * 6 + ["OverlappedGradReducer"] * 2, hooks


def test_single_rank_has_no_exchange():
    from nerfstudio_thermal_amd import trainer as T

    assert T._dp_guard(types.SimpleNamespace()) is None  # no process group: one rank, no exchange
    port = _free_port()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        assert T._dp_guard(types.SimpleNamespace()) is None  # a one-rank group
    finally:
        dist.destroy_process_group()
