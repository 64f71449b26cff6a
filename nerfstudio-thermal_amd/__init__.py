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
# Must be in the environment before the first HIP call of the process; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
