"""thermal-nerfacto volume-rendering hot path for MI355X (gfx950).

HIP kernels + C ABI live in `csrc/` (built into `libthermal_nerf_hip.so` next to this file);
the Python modules mirror the reference's Model/Field/Sampler/Renderer interface for that path.
There is no CPU fallback: every op raises if the HIP library is missing.
"""
__version__ = "0.1.0"
