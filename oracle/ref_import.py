"""Import shim for the REFERENCE implementation (test infrastructure only).

Used only in the build container (where /root/reference exists) by
oracle/make_golden.py to generate golden vectors from the reference's own
torch ("implementation='torch'") path, and by tests that cross-check the oracle
restatement against the live reference when it is present.  Nothing here ships
to the GPU box as a dependency: /root/reference does not exist there.

The stubs only satisfy *imports* of packages that are absent in this image
(jaxtyping, nerfacc, tyro, torchmetrics, cv2, viser, tensorboard); none of the
stubbed symbols is ever called on the thermal-nerfacto hot path
(SURVEY.md section 8c / Appendix A).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TN_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nerfstudio"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Sub:
    """subscriptable / constructible / callable no-op"""

    def __class_getitem__(cls, item):
        return cls

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


def install_stubs():
    if "jaxtyping" not in sys.modules:
        _mod("jaxtyping", Float=_Sub, Int=_Sub, Shaped=_Sub, Bool=_Sub, UInt8=_Sub, Num=_Sub)
    if "nerfacc" not in sys.modules:
        _mod("nerfacc", OccGridEstimator=_Sub, accumulate_along_rays=None, pack_info=None,
             render_weight_from_density=None)
    if "tyro" not in sys.modules:
        conf = _mod("tyro.conf", Suppress=_Sub, FlagConversionOff=_Sub, subcommand=lambda *a, **k: None)
        extras = _mod("tyro.extras", subcommand_type_from_defaults=lambda *a, **k: None,
                      set_accent_color=lambda *a, **k: None)
        _mod("tyro", conf=conf, extras=extras, cli=lambda *a, **k: None)
    if "torchmetrics" not in sys.modules:
        _mod("torchmetrics")
        _mod("torchmetrics.functional", structural_similarity_index_measure=lambda *a, **k: None)
        _mod("torchmetrics.image", PeakSignalNoiseRatio=_Sub)
        _mod("torchmetrics.image.lpip", LearnedPerceptualImagePatchSimilarity=_Sub)
    if "cv2" not in sys.modules:
        _mod("cv2")
    if "viser" not in sys.modules:
        _mod("viser", transforms=_mod("viser.transforms", SO3=_Sub, SE3=_Sub))
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        _mod("torch.utils.tensorboard", SummaryWriter=_Sub)


def import_reference():
    """Returns the reference's `nerfstudio` package (import side effects only)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import nerfstudio  # noqa: F401

    return nerfstudio
