"""CPU-side checks of the C-ABI boundary: the library builds, loads, exports every declared symbol, and
validates its arguments before touching the GPU."""
import ctypes
import os
import subprocess

import pytest

import diffusionspatialcontrol_amd as dsc
from diffusionspatialcontrol_amd import _lib, build as dsc_build


@pytest.fixture(scope="module")
def lib():
    dsc_build.build(verbose=False)
    return dsc.load_library()


def test_exports_every_declared_symbol(lib):
    names = _lib.declared_symbols()
    assert "dsc_region_xattn_fwd" in names and len(names) >= 6
    nm = subprocess.run(["nm", "-D", "--defined-only", dsc.lib_path()], capture_output=True, text=True).stdout
    for n in names:
        assert hasattr(lib, n), n
        assert f" T {n}" in nm, n
    assert set(_lib._SIGNATURES) == set(names), "every declared entry point needs a ctypes signature"


def test_version_and_target(lib):
    assert lib.dsc_abi_version() == 1
    assert lib.dsc_target_arch() == b"gfx950"
    assert lib.dsc_status_string(0) == b"ok"
    assert b"workspace" in lib.dsc_status_string(-3)


def test_code_object_is_gfx950_only():
    """No multi-arch fat binary, no compatibility targets: the bundle holds gfx950 code objects only."""
    import re
    blob = open(dsc.lib_path(), "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_self_attention_kernel_has_no_hidden_register_loads():
    """The flash self-attention kernel stages K / V by LDS-DMA: no inline-asm load with a register destination (the
    construct whose result the compiler could move or spill before it landed - the cause of the round-1 fault at three
    waves per SIMD), and no instantiation of it spills or uses scratch (checked on the built code object's metadata)."""
    import os
    import re
    import subprocess
    src = open(os.path.join(os.path.dirname(dsc.lib_path()), "csrc", "self_attn.hip")).read()
    for stmt in re.findall(r"asm\s*(?:volatile)?\s*\((.*?)\);", src, flags=re.S):
        assert not re.search(r"(global|buffer|flat|ds)_(load|read)", stmt), stmt      # only waits / v_max3 remain in asm
    assert "raw_ptr_buffer_load_lds" in src
    import tempfile
    root = os.path.dirname(os.path.dirname(dsc.lib_path()))
    with tempfile.TemporaryDirectory() as tmp:                   # device-only compile of the same source with the build's flags
        work = os.path.join(tmp, "self_attn.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only",
                               "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "diffusionspatialcontrol_amd", "csrc"),
                               "-S", os.path.join(root, "diffusionspatialcontrol_amd", "csrc", "self_attn.hip"), "-o", work],
                              stderr=subprocess.DEVNULL)
        notes = open(work).read()
        notes = notes[notes.index("amdhsa.kernels:"):]              # the code object's metadata (YAML) at the end of the listing
    kernels = re.findall(r"\.name:\s+(\S*self_attn_fwd\S*)(.*?)\.wavefront_size", notes, flags=re.S)
    assert len(kernels) >= 20
    for name, body in kernels:
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", body), name
        assert re.search(r"\.vgpr_spill_count:\s+0\b", body), name


def test_argument_validation_needs_no_gpu(lib):
    s3 = (ctypes.c_int64 * 3)(8 * 64 * 40, 320, 40)
    dummy = ctypes.c_void_p(0x1000)
    call = lambda **kw: lib.dsc_region_xattn_fwd(  # noqa: E731
        kw.get("q", dummy), dummy, dummy, dummy, kw.get("region", None), kw.get("Bc", 2), 8, 64, kw.get("S", 77),
        kw.get("d", 40), kw.get("Bw", 2), kw.get("ng", 1), s3, s3, s3, s3, 1.0, None, 0.0, kw.get("dtype", 0), 0,
        kw.get("ws", None), 0, None)
    assert call(q=None) == -1                      # null pointer
    assert call(Bc=0) == -1
    assert call(ng=3) == -1                        # groups must divide Bc
    assert call(d=44) == -2                        # head dim not a multiple of 8
    assert call(d=168) == -2
    assert call(S=97) == -2
    assert call(dtype=7) == -2
    assert call(q=ctypes.c_void_p(0x1004)) == -2   # misaligned
    assert call(region=dummy, Bw=3) == -1          # Bw must divide Bc*H
    assert call(region=dummy, ws=None) == -3       # statistics pass needs the workspace
    assert lib.dsc_region_xattn_workspace_bytes(2, 8, 4096, 77, 40, 1) >= 2 * 8 * 8 * 16
    assert lib.dsc_region_xattn_workspace_bytes(0, 8, 4096, 77, 40, 1) == 0


def test_ops_fail_loudly_without_gpu_or_library(monkeypatch):
    import torch
    from diffusionspatialcontrol_amd import ops
    q = torch.zeros(1, 1, 32, 8, dtype=torch.float16)
    with pytest.raises(dsc.DscLibraryError):
        ops.region_xattn(q, q, q)                  # CPU tensors: no fallback
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "lib_path", lambda: os.path.join("/nonexistent", "libdsc_hip.so"))
    with pytest.raises(dsc.DscLibraryError):
        _lib.load_library()


def test_workspace_slot_and_concat_groupnorm_validation(lib):
    """dsc_set_workspace_slot: four slots, thread-local; dsc_groupnorm_silu_nhwc_cat: argument checks before any launch"""
    import threading
    assert lib.dsc_set_workspace_slot(4) == -1 and lib.dsc_set_workspace_slot(-1) == -1
    assert lib.dsc_set_workspace_slot(3) == 0
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.dsc_set_workspace_slot(1)))       # another thread, its own slot
    t.start()
    t.join()
    assert seen == [0] and lib.dsc_set_workspace_slot(0) == 0
    d = ctypes.c_void_p(0x1000)
    cat = lambda **kw: lib.dsc_groupnorm_silu_nhwc_cat(  # noqa: E731
        d, kw.get("x2", ctypes.c_void_p(0x2000)), kw.get("C1", 320), kw.get("cat", ctypes.c_void_p(0x3000)),
        ctypes.c_void_p(0x4000), d, d, None, 0, 2, kw.get("C", 640), 64, 32, 1e-5, 1, 0, kw.get("ws", None), 0, None)
    assert cat(x2=None) == -1 and cat(cat=None) == -1
    assert cat(C1=640) == -1 and cat(C1=0) == -1          # both sources must contribute channels
    assert cat(C1=12) == -2                               # 16-byte vectors must not straddle the seam
    assert cat(cat=d) == -2                               # the concatenation must not alias a source
    assert cat() == -3                                    # valid arguments, no workspace


def test_tuning_profile_entry_points(lib):
    """dsc_set_tuning_profile / dsc_get_tuning_profile: two profiles, anything else is a bad argument and changes nothing"""
    assert lib.dsc_get_tuning_profile() == 0                                    # DSC_TUNE_LATENCY is the default
    assert lib.dsc_set_tuning_profile(1) == 0 and lib.dsc_get_tuning_profile() == 1
    assert lib.dsc_set_tuning_profile(2) == -1 and lib.dsc_set_tuning_profile(-1) == -1 and lib.dsc_get_tuning_profile() == 1
    assert lib.dsc_set_tuning_profile(0) == 0 and lib.dsc_get_tuning_profile() == 0


def test_groupnorm_partial_sums_need_two_channels_per_group(lib):
    """round-3 advisor: gn_tile_partials writes at most 32 group slots per 64-channel tile, so a grouping with ONE channel per
    group (C == groups == 64) would leave half the slots unwritten - both producers must decline it (0 rows -> the caller
    runs the two-launch GroupNorm), while 2..64 channels per group are covered.  Host-side planning only: no GPU."""
    assert lib.dsc_linear_gn_rows(8192, 64, 320, 4096, 64) == 0
    assert lib.dsc_linear_gn_rows(8192, 64, 320, 4096, 32) > 0
    assert lib.dsc_linear_gn_rows(8192, 320, 320, 4096, 32) > 0
    assert lib.dsc_conv3x3_gn_rows(2, 64, 64, 64, 64, 64, 0) == 0
    assert lib.dsc_conv3x3_gn_rows(2, 64, 64, 64, 64, 32, 0) > 0
    assert lib.dsc_conv3x3_gn_rows(2, 64, 64, 320, 320, 32, 0) > 0


def test_default_build_does_not_link_hipblaslt(lib):
    """the hipBLASLt fallback is a build option (DSC_WITH_HIPBLASLT=1): the default library neither links it nor claims it, and
    its entry point declines instead of launching"""
    if dsc_build.WITH_HIPBLASLT:
        pytest.skip("this tree was built with DSC_WITH_HIPBLASLT=1")
    assert lib.dsc_has_library_gemm() == 0
    needed = subprocess.run(["readelf", "-d", dsc.lib_path()], capture_output=True, text=True).stdout
    assert "hipblaslt" not in needed
    st3 = (ctypes.c_longlong * 3)(7, 7, 7)
    lib.dsc_linear_lt_stats(st3)
    assert list(st3) == [0, 0, 0]
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.dsc_linear_lt_f16(p, p, None, None, p, 8, 8, 8, 8, 8, 8, 0, None) != 0
