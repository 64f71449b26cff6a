"""thermal-nerfacto volume-rendering hot path for MI355X (gfx950).

HIP kernels + C ABI live in `csrc/` (built into `libthermal_nerf_hip.so` next to this file);
the Python modules mirror the reference's Model/Field/Sampler/Renderer interface for that path.
There is no CPU fallback: every op raises if the HIP library is missing.
"""
__version__ = "0.1.0"

# Hardware queues.  The ROCm runtime multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that
# share a queue serialise behind each other.  The package does NOT touch the environment (round 4 wrote GPU_MAX_HW_QUEUES=8 into it at import:
# process-global, silently ignored once HIP is initialised, and the optimum of one particular set of streams).  Instead the schedules fit the
# default: the single-GPU step uses four streams on proposal-update iterations (main, two proposal side streams, the d-position companion) and
# one otherwise; the data-parallel step uses three (main, ONE side stream for both proposal networks, RCCL's) -- profiles/r05_dp_hwq_sweep.json
# holds bench.py --force-dp for GPU_MAX_HW_QUEUES in {unset, 4, 5, 7, 8, 16}.  If a deployment adds streams of its own (more communicators, a
# prefetcher), GPU_MAX_HW_QUEUES=8 in the job's environment is the first thing to try, and parallel.ScheduleGuard falls back to the
# simple schedule in process when the overlapped one measures slower than it should.


def configure_host(single_thread_backward: bool = True) -> None:
    """OPT-IN process setting for a training process that drives the model through the reference Trainer's sequence (engine/trainer.py:455-499).
    Nothing in the package calls this at import; the method plugin calls it when TN_SINGLE_THREAD_BACKWARD=1 is in the environment.

    single_thread_backward: torch.autograd.set_multithreading_enabled(False) -- `loss.backward()` runs its nodes on the calling thread instead of
    handing them to autograd's device thread.  This model's backward is two custom nodes that only enqueue kernels, so the second thread buys
    nothing and the hand-over (wake-up, GIL traffic with the waiting caller) sits on the iteration's critical path: with the Trainer's two
    `grad_scaler.get_scale()` host synchronisations per iteration the GPU idles exactly as long as the host needs to reach the backward launch.
    Measured on MI355X, 4096 rays, autocast + GradScaler: 1.49 -> 1.27 ms per iteration (scripts/trace_api_host.py, ST_BWD=1)."""
    import torch

    torch.autograd.set_multithreading_enabled(not single_thread_backward)
