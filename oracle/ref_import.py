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


class _PermMeta(type):
    """class attributes of a permissive stub are permissive stubs (`mp.Process`, `cv2.INTER_AREA`, ...)"""

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Perm


class _Perm(metaclass=_PermMeta):
    """subscriptable / constructible / callable / subclassable no-op whose every attribute is another one"""

    def __class_getitem__(cls, item):
        return cls

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Perm()


def _perm_mod(name, **attrs):
    def _getattr(attr):
        if attr.startswith("__"):  # inspect.getmodule walks sys.modules reading __file__: a stub must not answer that
            raise AttributeError(attr)
        return _Perm

    m = types.ModuleType(name)
    m.__getattr__ = _getattr
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# Third-party packages the reference's TRAINER / method table import at module level and this image lacks (found by importing
# nerfstudio.configs.method_configs here and reading each ModuleNotFoundError).  None of them is on the thermal-nerfacto path: they belong to
# other dataparsers (imageio, newrawpy, pyquaternion, splines), the viewer (viser.*), splatfacto (gsplat, pytorch_msssim), the parallel
# datamanager (pathos) and the CLI (tyro, appdirs).
EXTENDED_ABSENT = ("imageio", "imageio.v3", "appdirs", "newrawpy", "pyquaternion", "pathos", "pathos.helpers", "splines", "splines.quaternion",
                   "gsplat", "gsplat._torch_impl", "gsplat.project_gaussians", "gsplat.rasterize", "gsplat.sh", "pytorch_msssim",
                   "viser.theme", "viser.infra")


def install_extended_stubs():
    """The 7 stubs of install_stubs() are enough for the MODEL path (what the goldens are generated from).  The reference's Trainer, its method
    table (configs/method_configs.py) and datamanagers import more absent third-party packages at module level; for the tests that mix
    trainer.FusedTrainerMixin into the real Trainer and build the method plugin (tests/test_real_trainer_cpu.py) those get PERMISSIVE stubs --
    any attribute is a subscriptable / subclassable no-op.  Build container only; nothing here ships."""
    install_stubs()
    import importlib.util

    for name in ("cv2", "viser", "viser.transforms", "tyro", "tyro.conf", "tyro.extras"):  # the import-only stubs above, made permissive
        if not isinstance(sys.modules.get(name), types.ModuleType) or getattr(sys.modules[name], "__file__", None) is None:
            sys.modules.pop(name, None)
    conf = _perm_mod("tyro.conf", subcommand=lambda *a, **k: None)
    extras = _perm_mod("tyro.extras", subcommand_type_from_defaults=lambda *a, **k: None, set_accent_color=lambda *a, **k: None)
    _perm_mod("tyro", conf=conf, extras=extras, cli=lambda *a, **k: None)
    _perm_mod("cv2")
    _perm_mod("viser", transforms=_perm_mod("viser.transforms"))
    for name in EXTENDED_ABSENT:
        top = name.split(".")[0]
        if name in sys.modules:
            continue
        try:
            real = top not in sys.modules and importlib.util.find_spec(top) is not None
        except (ImportError, ValueError):
            real = False
        if not real:
            _perm_mod(name)


def import_reference_trainer():
    """-> (Trainer, TrainerConfig, method_configs, MethodSpecification) of the reference, imported with the extended stubs."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_extended_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from nerfstudio.configs.method_configs import method_configs
    from nerfstudio.engine.trainer import Trainer, TrainerConfig
    from nerfstudio.plugins.types import MethodSpecification

    return Trainer, TrainerConfig, method_configs, MethodSpecification


def import_reference():
    """Returns the reference's `nerfstudio` package (import side effects only)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import nerfstudio  # noqa: F401

    return nerfstudio
