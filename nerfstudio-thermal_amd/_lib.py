"""ctypes binding of libthermal_nerf_hip.so (the C ABI declared in include/thermal_nerf_hip.h).

There is deliberately no fallback: if the HIP library is missing the first op raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# TN_LIB names another build of the same library (A/B timing of kernel variants); there is still no fallback if it cannot be loaded
LIB_PATH = os.path.abspath(os.environ["TN_LIB"]) if os.environ.get("TN_LIB") else os.path.join(_HERE, "libthermal_nerf_hip.so")
ABI_VERSION = 307  # include/thermal_nerf_hip.h as this binding was written for (tn_version() of the library must match)
TN_MAX_LEVELS = 16
TN_MAX_SAMPLES = 256
TN_RENDER_SCRATCH_FLOATS = 4096
TN_LOSS_LINES = 64
TN_RENDER_TRAIN_OFFSETS = 25
TN_BWD_MLP, TN_BWD_SCATTER, TN_BWD_JOIN, TN_BWD_SCATTER_BIN, TN_BWD_SCATTER_FOLD, TN_BWD_FORK_DPOS, TN_BWD_COUNTERS_CLEAN = 1, 2, 4, 8, 16, 32, 64

_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f = C.c_float
_d = C.c_double


class TnGrid(C.Structure):
    _fields_ = [
        ("table", _p),
        ("table_grad", _p),
        ("num_levels", _i32),
        ("log2_hashmap_size", _i32),
        ("res", _f * TN_MAX_LEVELS),
        ("nonfinite_flag", _p),
        ("table_grad_is_zero", _i32),
    ]


class TnPropNet(C.Structure):
    _fields_ = [("grid", TnGrid)] + [(n, _p) for n in ("w0", "b0", "w1", "b1", "gw0", "gb0", "gw1", "gb1")]


_FIELD_PTRS = ("w0", "b0", "w1", "b1", "hw0", "hb0", "hw1", "hb1", "hw2", "hb2", "emb",
               "gw0", "gb0", "gw1", "gb1", "ghw0", "ghb0", "ghw1", "ghb1", "ghw2", "ghb2", "gemb")


class TnField(C.Structure):
    _fields_ = [("grid", TnGrid)] + [(n, _p) for n in _FIELD_PTRS] + [("num_channels", _i32), ("num_images", _i32)]


TN_TRAIN_STEP_MAX_RANGES = 8
_R = TN_TRAIN_STEP_MAX_RANGES


class TnSampleRays(C.Structure):
    """include/thermal_nerf_hip.h: the arguments of tn_sample_rays as a block (TnTrainStep.next_sample)"""
    _fields_ = [("images", _p), ("image_offsets", _p), ("heights", _p), ("widths", _p), ("is_thermal", _p), ("image_idx", _p), ("num_images", _i32),
                ("u", _p), ("num_rays", _i64), ("patch_size", _i32), ("ray_indices", _p), ("image", _p), ("is_thermal_out", _p), ("camera_indices", _p),
                ("c2w", _p), ("fx", _p), ("fy", _p), ("cx", _p), ("cy", _p), ("distortion", _p), ("num_cameras", _i32), ("origins", _p),
                ("directions", _p), ("pixel_area", _p), ("directions_norm", _p)]


class TnNextSampling(C.Structure):
    """include/thermal_nerf_hip.h: the NEXT iteration's sampling front as co-work of this iteration's optimiser launch (TnTrainStep.next_sampling)"""
    _fields_ = [("fwd_out", _p), ("jitter0", _p), ("jitter1", _p), ("jitter2", _p), ("anneal", _f), ("prop_grad", _i32)]


class TnTrainStep(C.Structure):
    """The argument block of tn_train_step: field for field the struct of include/thermal_nerf_hip.h (tests/test_abi_cpu.py compares the two)."""
    _fields_ = [
        ("prop0", C.POINTER(TnPropNet)), ("prop1", C.POINTER(TnPropNet)), ("field", C.POINTER(TnField)),
        ("origins_in", _p), ("directions_in", _p), ("camera_indices", _p), ("image", _p), ("is_thermal", _p), ("nears", _p), ("fars", _p),
        ("N", _i64), ("S0", _i32), ("S1", _i32), ("S2", _i32),
        ("pose_adjustment", _p), ("frozen", _p), ("num_cameras", _i32), ("grad_pose", _p),
        ("trans_pen", _f), ("rot_pen", _f), ("pen_scale", _f),
        ("anneal", _f), ("prop_grad", _i32),
        ("jitter0", _p), ("jitter1", _p), ("jitter2", _p), ("lin_spaced0", _p), ("lin_pdf1", _p), ("lin_pdf2", _p),
        ("field_workspace", _p), ("field_workspace_bytes", _i64),
        ("prop_workspace0", _p), ("prop_workspace_bytes0", _i64), ("prop_workspace1", _p), ("prop_workspace_bytes1", _i64),
        ("fwd_out", _p), ("bwd_tmp", _p), ("acc", _p), ("acc_bytes", _i64),
        ("losses16", _p), ("loss_lines", _p), ("d_comp", _p), ("d_weights0", _p), ("d_weights1", _p), ("d_weights2", _p), ("d_origins", _p),
        ("d_directions", _p),
        ("thermal_mult", _f), ("tv_mult", _f), ("cross_mult", _f), ("distortion_mult", _f), ("interlevel_mult", _f),
        ("num_check", _i32), ("check_offsets", _i64 * _R), ("check_counts", _i64 * _R), ("check_flags", _i32 * _R), ("pose_flag", _i32),
        ("params", _p), ("grads", _p), ("exp_avg", _p), ("exp_avg_sq", _p),
        ("num_ranges", _i32), ("offsets", _i64 * _R), ("counts", _i64 * _R), ("steps", _i32 * _R),
        ("lrs", _d * _R), ("lr_finals", _d * _R), ("sched_max_steps", _i32 * _R), ("flag_index", _i32 * _R), ("sched_step", _i32),
        ("beta1", _d), ("beta2", _d), ("eps", _d),
        ("found_inf", _p), ("num_flags", _i32), ("skipped", _p), ("lag_index", _i32),
        ("scale", _p), ("growth_tracker", _p), ("done_counter", _p), ("growth_factor", _d), ("backoff_factor", _d), ("growth_interval", _i32),
        ("next_sample", C.POINTER(TnSampleRays)), ("next_sample_taken", C.POINTER(_i32)),
        ("next_sampling", C.POINTER(TnNextSampling)), ("next_sampling_taken", C.POINTER(_i32)), ("sampling_done", _i32),
    ]


class TnSplatCamera(C.Structure):
    _fields_ = [("viewmat", _f * 12), ("projmat", _f * 16), ("fx", _f), ("fy", _f), ("cx", _f), ("cy", _f), ("position", _f * 3),
                ("clip_thresh", _f), ("width", _i32), ("height", _i32)]


# name -> (restype, argtypes); must list every symbol include/thermal_nerf_hip.h declares
SIGNATURES = {
    "tn_last_error": (C.c_char_p, []),
    "tn_version": (C.c_int, []),
    "tn_field_workspace_bytes": (_i64, [_i64, _i32]),
    "tn_field_encode_plan": (_i32, [_p, _i64, _p]),
    "tn_prop_workspace_bytes": (_i64, [_i64]),
    "tn_sample_pixels": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "tn_raygen": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _i64, _p, _p, _p, _p, _p]),
    "tn_sample_rays": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "tn_pose_spaced_bins": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p]),
    "tn_pose_bwd_finish": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _f, _f, _f, _p, _p]),
    "tn_pose_bwd_finish_check": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _f, _f, _f, _p, _p, _i32, _p, _p, _p, _i32, _p, _i32, _p]),
    "tn_pose_apply_fwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p]),
    "tn_pose_apply_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "tn_spaced_bins": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p, _p]),
    "tn_prop_density_fwd": (C.c_int, [C.POINTER(TnPropNet), _p, _p, _p, _i64, _i32, _p, _p]),
    "tn_prop_density_bwd": (C.c_int, [C.POINTER(TnPropNet), _p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p, _p]),
    "tn_hash_scatter_workspace_bytes": (_i64, [_i64, _i32]),
    "tn_hash_scatter": (C.c_int, [C.POINTER(TnGrid), _p, _p, _p, _p, _i32, _i64, _i32, _p, _p, _p, _i64, _p]),
    "tn_weights_fwd": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "tn_weights_bwd": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p]),
    "tn_weights_resample": (C.c_int, [_p, _p, _p, _i32, _f, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "tn_pdf_resample": (C.c_int, [_p, _p, _i32, _f, _p, _p, _p, _p, _i64, _i32, _p, _p, _p]),
    "tn_field_pack_weights": (C.c_int, [C.POINTER(TnField), _p, _p]),
    "tn_field_fwd": (C.c_int, [C.POINTER(TnField), _p, _p, _p, _p, _i64, _i32, _i32, _p, _i64, _p, _p, _p, _p]),
    "tn_field_bwd": (C.c_int, [C.POINTER(TnField), _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p, _p]),
    "tn_field_bwd_phase": (C.c_int, [C.POINTER(TnField), _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p, _i32, _i32, _i32, _p]),
    "tn_field_dense_count": (_i64, [C.POINTER(TnField), _i64, _i32, _i32]),
    "tn_field_bwd_scatter_dense": (C.c_int, [C.POINTER(TnField), _p, _p, _p, _i64, _i32, _p, _i64, _p, _p, _i32, _i32, _p, _p]),
    "tn_field_dense_fold": (C.c_int, [C.POINTER(TnField), _i64, _i32, _i32, _p, _p]),
    "tn_field_density_fwd": (C.c_int, [C.POINTER(TnField), _p, _p, _p, _i64, _i32, _i32, _p, _i64, _p, _p]),
    "tn_minmax_init": (C.c_int, [_p, _p]),
    "tn_composite_fwd": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    "tn_clip_depth": (C.c_int, [_p, _p, _i64, _p]),
    "tn_composite_bwd": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "tn_render_rays_eval_workspace_bytes": (_i64, [_i64, _i32, _i32, _i32, _i32]),
    "tn_render_rays_eval": (C.c_int, [C.POINTER(TnPropNet), C.POINTER(TnPropNet), C.POINTER(TnField), _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _f,
                                      _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "tn_render_rays_train_layout": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _p, _i32]),
    "tn_render_rays_train": (C.c_int, [C.POINTER(TnPropNet), C.POINTER(TnPropNet), C.POINTER(TnField), _p, _p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _i32,
                                       _i32, _f, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _i32, _p]),
    "tn_render_fwd": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "tn_render_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "tn_distortion_loss": (C.c_int, [_p, _p, _i64, _i32, _f, _p, _p, _p]),
    "tn_interlevel_loss": (C.c_int, [_p, _p, _i32, _p, _p, _i32, _i64, _f, _p, _p, _p]),
    "tn_proposal_losses": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _p, _p, _i64, _f, _f, _p, _p, _p, _p]),
    "tn_pixel_losses": (C.c_int, [_p, _i32, _p, _i32, _p, _p, _i64, _f, _f, _f, _p, _p, _p, _p]),
    "tn_sample_rays_args": (C.c_int, [C.POINTER(TnSampleRays), _p]),
    "tn_render_losses_bwd": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _f, _f, _p, _p, _p, _f, _f, _f, _p, _p, _p, _p,
                                       _i32, _p]),
    "tn_train_losses": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _p, _p, _i64, _f, _f, _p, _p, _i32, _p, _i32, _p, _p, _f, _f, _f, _p, _p, _p, _p]),
    "tn_losses_finish": (C.c_int, [_p, _p, _p, _i32, _f, _f, _f, _p, _p, _p]),
    "tn_l1_loss": (C.c_int, [_p, _p, _i64, _f, _f, _p, _p, _p, _p]),
    "tn_camera_reg": (C.c_int, [_p, _i32, _f, _f, _f, _p, _p, _p]),
    "tn_train_metrics": (C.c_int, [_p, _i64, _f, _p, _i32, _p, _i32, _p, _p]),
    "tn_adam_step": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _d, _d, _d, _d, _p]),
    "tn_adam_step_ranges": (C.c_int, [_p, _p, _p, _p, _i32, _p, _p, _p, _p, _d, _d, _d, _p]),
    "tn_grad_nonfinite": (C.c_int, [_p, _i64, _p, _p]),
    "tn_adam_step_ranges_amp": (C.c_int, [_p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _i32, _d, _d, _d, _p, _p, _p, _i32, _p, _i32, _i32, _i32, _p]),
    "tn_grad_nonfinite_ranges": (C.c_int, [_p, _i32, _p, _p, _p, _i32, _p, _p]),
    "tn_grad_scaler_update": (C.c_int, [_p, _p, _p, _i32, _p, _d, _d, _i32, _i32, _p]),
    "tn_adam_step_ranges_amp_update": (C.c_int, [_p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _i32, _d, _d, _d, _p, _p, _p, _i32, _p, _i32, _i32, _i32,
                                                 _p, _p, _p, _d, _d, _i32, _p]),
    "tn_fill_zero": (C.c_int, [_p, _i64, _p]),
    "tn_train_step": (C.c_int, [C.POINTER(TnTrainStep), _p]),
    "tn_shutdown": (C.c_int, []),
    "tn_comm_unique_id": (C.c_int, [_p]),
    "tn_comm_create": (C.c_int, [_p, _i32, _i32, C.POINTER(_p)]),
    "tn_comm_destroy": (C.c_int, [_p]),
    "tn_allreduce_grads": (C.c_int, [_p, _p, _i64, _i32, _p]),
    "tn_render_rays_train_bwd_tmp_floats": (_i64, [_i64, _i32, _i32, _i32, _i32]),
    "tn_render_rays_train_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32] + [_p] * 7 + [_i64, _p, _i64, _p, _i64] + [_p] * 3 + [_i32, _p]),
    "tn_splat_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "tn_splat_project": (C.c_int, [_p] * 9 + [_i64, _i32, _i32, _i32] + [_p] * 8 + [_i64, _p]),
    "tn_splat_bin": (C.c_int, [_p, _p, _i64, _p, _i64, _p, _p]),
    "tn_splat_raster": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _p, _p, _p, _p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load (once) and return the HIP library; raise loudly when it is not built."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; /opt/rocm ships another with the SAME soname.  Whichever is loaded first serves the whole
        # process, and a process that ends up with ROCm's runtime under torch's HSA loses the device ("no ROCm-capable device").  Importing
        # torch first makes its runtime the one this library binds to, so kernels launched here and by torch share streams and memory.
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the thermal-nerfacto hot path."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.tn_version() != ABI_VERSION:  # signatures are positional: a stale build would take a workspace pointer for a flag
            raise RuntimeError(f"{LIB_PATH} reports ABI version {lib.tn_version()}, this binding needs {ABI_VERSION}: rebuild it (__graft_entry__.build())")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().tn_last_error()
        raise RuntimeError(f"libthermal_nerf_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")
