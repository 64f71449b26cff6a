"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports exactly what the header declares."""
import os
import re

import pytest

import nerfstudio_thermal_amd  # noqa: F401
from nerfstudio_thermal_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _lib.load()


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "thermal_nerf_hip.h")).read()
    return set(re.findall(r"\b(tn_[a-z0-9_]+)\s*\(", hdr))


def test_header_and_binding_agree():
    assert header_symbols() == set(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.tn_version() >= 100


def test_host_side_argument_validation_without_gpu(lib):
    # shape/pointer validation happens before any launch, so it can be exercised without a device
    assert lib.tn_spaced_bins(None, None, None, None, 4, 16, None, None, None) == -22
    assert b"null pointer" in lib.tn_last_error()
    assert lib.tn_field_workspace_bytes(-1, 0) == -22
    assert lib.tn_field_workspace_bytes(4096 * 48, 1) > lib.tn_field_workspace_bytes(4096 * 48, 0) > 0
    assert lib.tn_prop_workspace_bytes(1024) >= 1024 * 49 * 4


def test_ops_refuse_cpu_tensors():
    import torch

    from nerfstudio_thermal_amd import ops

    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.spaced_bins(torch.zeros(4, 1), torch.ones(4, 1), 16)


def test_level_resolutions_match_reference_probe():
    from nerfstudio_thermal_amd import ops

    assert ops.level_resolutions(16, 16, 2048) == [16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]
    assert ops.level_resolutions(5, 16, 128) == [16, 26, 45, 76, 128]
    assert ops.level_resolutions(5, 16, 256) == [16, 32, 64, 128, 256]


@pytest.mark.parametrize("points", [1, 511, 512, 3072, 4096 * 48, 640 * 480 // 4])
@pytest.mark.parametrize("levels,log2t", [(16, 19), (16, 12), (5, 17), (1, 19), (9, 19)])
def test_encode_plan_covers_every_level_chunk_once(lib, points, levels, log2t):
    """The XCD-affine gather's work plan (host logic, no GPU): every chunk of every level appears exactly once, and an XCD only sees levels x, x + 8."""
    import ctypes as C

    from nerfstudio_thermal_amd import ops

    g = _lib.TnGrid()
    g.num_levels, g.log2_hashmap_size = levels, log2t
    for i, r in enumerate(ops.level_resolutions(levels, 16, 2048)):
        g.res[i] = float(r)
    out = (C.c_int32 * (8 * 2 * 3))()
    chunks = lib.tn_field_encode_plan(C.byref(g), points, out)
    assert chunks == -(-points // 512)
    covered = {l: 0 for l in range(levels)}
    for x in range(8):
        for i in range(2):
            l, c0, cnt = out[(x * 2 + i) * 3: (x * 2 + i) * 3 + 3]
            if l < 0:
                assert cnt == 0
                continue
            assert l % 8 == x and c0 == covered[l] and cnt > 0
            covered[l] += cnt
    assert all(v == chunks for v in covered.values())
    assert lib.tn_field_encode_plan(None, points, out) == -22


def test_short_workspaces_are_refused_before_any_launch(lib):
    """ABI 302: every entry point that writes a workspace takes its size and answers TN_EINVAL to a short buffer (the scratch of a backward
    pass is hundreds of MB: the callee used to trust the pointer).  Validation happens on the host before the first launch, so it runs here
    without a device; the pointers only have to be non-NULL."""
    import ctypes as C

    from nerfstudio_thermal_amd import ops

    buf = (C.c_float * 64)()
    ptr = C.cast(buf, C.c_void_p)
    fake = C.c_void_p(256 * 4096)  # a 256-byte aligned, non-NULL address that is never dereferenced
    N, S = 64, 48
    g = _lib.TnGrid()
    g.table, g.table_grad, g.num_levels, g.log2_hashmap_size = ptr, ptr, 16, 19
    for i, r in enumerate(ops.level_resolutions(16, 16, 2048)):
        g.res[i] = float(r)
    need = lib.tn_hash_scatter_workspace_bytes(N * S, 16)
    assert need > 0
    assert lib.tn_hash_scatter(C.byref(g), ptr, ptr, ptr, ptr, -1, N, S, None, None, fake, need - 1, None) == -22
    assert b"workspace of" in lib.tn_last_error()
    f = _lib.TnField()
    f.grid = g
    for name in _lib._FIELD_PTRS:
        setattr(f, name, ptr)
    f.num_channels, f.num_images = 4, 8
    need = lib.tn_field_workspace_bytes(N * S, 1)
    assert lib.tn_field_bwd(C.byref(f), ptr, ptr, ptr, ptr, ptr, ptr, N, S, fake, need - 1, None, None, None) == -22
    assert b"tn_field_workspace_bytes" in lib.tn_last_error()
    assert lib.tn_field_fwd(C.byref(f), ptr, ptr, ptr, ptr, N, S, 1, fake, 16, ptr, ptr, None, None) == -22
    assert lib.tn_field_density_fwd(C.byref(f), ptr, ptr, ptr, N, S, 1, fake, 16, ptr, None) == -22
    p = _lib.TnPropNet()
    p.grid = g
    p.grid.num_levels, p.grid.log2_hashmap_size = 5, 17
    for name in ("w0", "b0", "w1", "b1", "gw0", "gb0", "gw1", "gb1"):
        setattr(p, name, ptr)
    need = lib.tn_prop_workspace_bytes(N * 256)
    assert lib.tn_prop_density_bwd(C.byref(p), ptr, ptr, ptr, ptr, N, 256, fake, need - 1, None, None, None) == -22
    assert b"tn_prop_workspace_bytes" in lib.tn_last_error()
    need = lib.tn_render_rays_eval_workspace_bytes(N, 256, 96, 48, 4)
    assert lib.tn_render_rays_eval(C.byref(p), C.byref(p), C.byref(f), ptr, ptr, ptr, ptr, ptr, N, 256, 96, 48, 1.0, ptr, ptr, ptr, fake, need - 1,
                                   ptr, None, None, None, None, None, ptr, None, None, None) == -22


def test_field_camera_cap_is_the_documented_one():
    """include/thermal_nerf_hip.h documents TN_FIELD_MAX_IMAGES (cameras the backward's per-camera sums are sized for); the kernel source sizes
    the workspace region with its own constant: they must be the same number."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "thermal_nerf_hip.h")).read()
    src = open(os.path.join(root, "nerfstudio-thermal_amd", "csrc", "tn_field.hip")).read()
    a = int(re.search(r"#define TN_FIELD_MAX_IMAGES (\d+)", hdr).group(1))
    b = int(re.search(r"#define FIELD_MAX_IMAGES (\d+)", src).group(1))
    assert a == b == 4096


def test_library_exports_exactly_the_header(lib):
    """`nm -D` of the built library = the header's entry points, nothing else: the library is linked with a version script (csrc/exports.map) and
    compiled with -fvisibility=hidden, so kernel handles and the helpers its translation units share stay local."""
    import shutil
    import subprocess

    if shutil.which("nm") is None:
        pytest.skip("no nm")
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported == header_symbols(), exported ^ header_symbols()


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of the header's structs against the C compiler's layout: size and the offset of every
    field (a field added on one side only, or in another order, would hand tn_train_step pointers in the wrong slots)."""
    import ctypes as C
    import shutil
    import subprocess

    from nerfstudio_thermal_amd import _lib as L

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "thermal_nerf_hip.h")).read()
    # (TnSampleRays / TnNextSampling: the next iteration's batch and sampling front, TnTrainStep::next_sample / next_sampling; the others: every
    # struct an entry point takes by pointer)
    for struct in ("TnTrainStep", "TnSampleRays", "TnNextSampling", "TnGrid", "TnPropNet", "TnField", "TnSplatCamera"):
        mirror = getattr(L, struct)
        names = [f[0] for f in mirror._fields_]
        src = tmp_path / f"layout_{struct}.c"
        src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) {\n  printf("%%zu\\n", sizeof(%s));\n%s  return 0;\n}\n'
                       % (os.path.join(root, "include", "thermal_nerf_hip.h"), struct,
                          "".join('  printf("%s %%zu\\n", offsetof(%s, %s));\n' % (n, struct, n) for n in names)))
        exe = tmp_path / f"layout_{struct}"
        subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
        out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
        assert int(out[0]) == C.sizeof(mirror), struct
        seen = {}
        for line in out[1:]:
            if line:
                n, o = line.split()
                seen[n] = int(o)
        assert seen == {n: getattr(mirror, n).offset for n in names}, struct
        # and the header declares no field the mirror lacks
        body = hdr[hdr.index("typedef struct %s {" % struct):hdr.index("} %s;" % struct)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        decl = set()
        for stmt in body.split(";"):
            stmt = stmt.replace("typedef struct %s {" % struct, "").strip()
            if not stmt:
                continue
            for part in stmt.split(","):
                m = re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", part.strip())
                decl.add(m.group(1))
        assert decl == set(names), (struct, decl ^ set(names))


def test_train_step_argument_block_is_validated_before_any_launch(lib):
    """tn_train_step refuses a missing or half-filled argument block on the host (no device needed) and accepts an empty batch."""
    import ctypes as C

    from nerfstudio_thermal_amd import _lib as L

    assert lib.tn_train_step(None, None) == -22
    assert b"null argument block" in lib.tn_last_error()
    a = L.TnTrainStep()  # all zero: N == 0, an empty batch touches nothing
    assert lib.tn_train_step(C.byref(a), None) == 0
    a.N = 4096
    assert lib.tn_train_step(C.byref(a), None) == -22
    assert b"null pointer" in lib.tn_last_error()


def test_sample_rays_block_is_validated_on_the_host(lib):
    """tn_sample_rays_args (the TnSampleRays form of tn_sample_rays; the block tn_train_step takes as next_sample): a missing block is refused, an
    empty batch touches nothing, a block without its arrays is refused before any launch."""
    import ctypes as C

    from nerfstudio_thermal_amd import _lib as L

    assert lib.tn_sample_rays_args(None, None) == -22
    assert b"null pointer" in lib.tn_last_error()
    a = L.TnSampleRays()
    assert lib.tn_sample_rays_args(C.byref(a), None) == 0  # num_rays == 0
    a.num_rays, a.num_images, a.patch_size = 64, 4, 2
    assert lib.tn_sample_rays_args(C.byref(a), None) == -22
    assert b"null pointer" in lib.tn_last_error()



def test_comm_entry_points_validate_on_the_host(lib):
    """tn_comm_* / tn_allreduce_grads (the exchange for a host that binds only the C ABI): argument validation needs neither a GPU nor RCCL."""
    import ctypes as C

    assert lib.tn_allreduce_grads(None, None, 16, 1, None) == -22
    assert b"bad argument" in lib.tn_last_error()
    assert lib.tn_comm_unique_id(None) == -22
    h = C.c_void_p()
    buf = C.create_string_buffer(128)
    assert lib.tn_comm_create(buf, 2, 5, C.byref(h)) == -22  # rank outside the world
    assert lib.tn_comm_create(None, 1, 0, C.byref(h)) == -22
    assert lib.tn_comm_destroy(None) == 0  # nothing to destroy
