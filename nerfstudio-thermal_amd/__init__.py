"""thermal-nerfacto volume-rendering hot path for MI355X (gfx950).

HIP kernels + C ABI live in `csrc/` (built into `libthermal_nerf_hip.so` next to this file);
the Python modules mirror the reference's Model/Field/Sampler/Renderer interface for that path.
There is no CPU fallback: every op raises if the HIP library is missing.
"""
__version__ = "0.1.0"

import os as _os

# The data-parallel step keeps six HIP streams busy (main, proposal side stream, the library's companion stream for the weight-gradient
# GEMMs, two RCCL communicators, copies).  The ROCm runtime multiplexes a process's streams onto 4 hardware queues by default; a stream
# that shares a queue with another one serialises behind it -- e.g. an all-reduce waiting for the proposal backward blocks the main
# stream's kernels queued after it (bench.py --force-dp: 2.75 ms/step with 4 queues, 1.42 ms with 8; the single-GPU step does not care).
# Round 4 sweep of the one-rank data-parallel step (profiles/r04_experiments.md): 8 -> 0.887 ms, 6 / 12 / 16 / 24 -> ~1.02, 4 / 5 / 7 -> 1.8-2.1 (the GPU idles
# ~0.9 ms per step), 2 / 3 -> 0.94-1.01; the single-GPU step is the same with 3, 4 and 8.  The runtime deals streams onto the queues in creation order, so
# the best count belongs to THIS set of streams: with another one (more communicators, another torch version) re-measure.
# Must be in the environment before the first HIP call of the process; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


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
