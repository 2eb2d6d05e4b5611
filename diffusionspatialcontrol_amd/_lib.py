"""ctypes binding of libdsc_hip.so (include/dsc_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DscLibraryError(RuntimeError):
    pass


def lib_path():
    # DSC_LIB_PATH: A/B measurements against another build of the same ABI (tools/, never the tests)
    return os.environ.get("DSC_LIB_PATH") or os.path.join(_HERE, "libdsc_hip.so")


def header_path():
    return os.path.join(os.path.dirname(_HERE), "include", "dsc_hip.h")


def declared_symbols():
    """Every function name declared in include/dsc_hip.h."""
    txt = open(header_path()).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dsc_[a-z0-9_]+)\s*\(", txt)))


_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p

_SIGNATURES = {
    "dsc_abi_version": (ctypes.c_int, []),
    "dsc_target_arch": (ctypes.c_char_p, []),
    "dsc_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "dsc_debug_set_stamp_buffer": (None, [_vp]),
    "dsc_debug_set_self_attn_variant": (None, [ctypes.c_int]),
    "dsc_debug_set_self_attn_stamps": (None, [_vp]),
    "dsc_debug_set_gemm_stamps": (None, [_vp]),
    "dsc_debug_set_self_attn_stamp_wave": (None, [ctypes.c_int]),
    "dsc_region_xattn_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "dsc_region_xattn_fwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp] + [ctypes.c_int] * 7 + [_i64p] * 4 +
                             [ctypes.c_float, _vp, ctypes.c_float, ctypes.c_int, ctypes.c_uint, _vp, ctypes.c_size_t, _vp]),
    "dsc_region_xattn_std": (ctypes.c_int, [_vp, _vp] + [ctypes.c_int] * 6 + [_i64p] * 2 +
                             [ctypes.c_float, ctypes.c_int, ctypes.c_uint, _vp, _vp, ctypes.c_size_t, _vp]),
    "dsc_region_xattn_std_masked": (ctypes.c_int, [_vp, _vp] + [ctypes.c_int] * 6 + [_i64p] * 2 +
                                    [ctypes.c_float, ctypes.c_int, ctypes.c_uint, _vp, _i64p, _vp, _vp, ctypes.c_size_t, _vp]),
    "dsc_xattn_kv_pack_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "dsc_xattn_kv_pack": (ctypes.c_int, [_vp] * 3 + [ctypes.c_int] * 4 + [_i64p] * 2 + [ctypes.c_int, _vp]),
    "dsc_region_xattn_fwd_packed": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int] * 8 + [_i64p] * 2 +
                                    [ctypes.c_float, _vp, ctypes.c_float, ctypes.c_int, ctypes.c_uint, _vp,
                                     ctypes.c_size_t, _vp]),
    "dsc_self_attn_fwd": (ctypes.c_int, [_vp] * 4 + [ctypes.c_int] * 5 + [_i64p] * 4 + [ctypes.c_float, ctypes.c_int, _vp]),
    "dsc_prepare_unet_input": (ctypes.c_int, [_vp] + [ctypes.c_float] * 3 + [_vp, _vp, _vp] + [ctypes.c_int] * 3 +
                               [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_cfg_dpmpp2m_step": (ctypes.c_int, [_vp, _vp, _vp] + [ctypes.c_float] * 8 + [_vp, _vp, _vp] +
                             [ctypes.c_int] * 3 + [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_dpmpp2m_update": (ctypes.c_int, [_vp, _vp, _vp] + [ctypes.c_float] * 3 + [_vp, ctypes.c_int64, ctypes.c_int, _vp]),
    "dsc_groupnorm_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "dsc_groupnorm_silu": (ctypes.c_int, [_vp] * 4 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                                             _vp, ctypes.c_size_t, _vp]),
    "dsc_groupnorm_nhwc_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "dsc_debug_set_gn_mode": (None, [ctypes.c_int]),
    "dsc_groupnorm_silu_nhwc": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int64] + [ctypes.c_int] * 4 +
                                [ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "dsc_groupnorm_silu_nhwc_cat": (ctypes.c_int, [_vp, _vp, ctypes.c_int] + [_vp] * 5 + [ctypes.c_int64] + [ctypes.c_int] * 4 +
                                    [ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "dsc_set_workspace_slot": (ctypes.c_int, [ctypes.c_int]),
    "dsc_set_tuning_profile": (ctypes.c_int, [ctypes.c_int]),
    "dsc_get_tuning_profile": (ctypes.c_int, []),
    "dsc_linear_lt_stats": (None, [_vp]),
    "dsc_has_library_gemm": (ctypes.c_int, []),
    "dsc_add_bias_residual": (ctypes.c_int, [_vp] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_linear_f16": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 3 +
                       [ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_debug_set_conv_stamps": (None, [_vp]),
    "dsc_debug_set_conv_ring": (None, [ctypes.c_int]),
    "dsc_conv3x3_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "dsc_conv3x3_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "dsc_conv3x3_nhwc_f16": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int] * 5 + [ctypes.c_int64] * 3 +
                             [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "dsc_conv3x3_fewcin_f16": (ctypes.c_int, [_vp] * 4 + [ctypes.c_int] * 6 + [_vp]),
    "dsc_linear_lt_f16": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 3 +
                          [ctypes.c_int, _vp]),
    "dsc_linear_rows_f16": (ctypes.c_int, [_vp] * 4 + [ctypes.c_int] * 3 + [ctypes.c_int64] * 2 + [ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_linear_ln_f16": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 3 +
                          [ctypes.c_int, _vp, ctypes.c_int, _vp, ctypes.c_float, _vp, ctypes.c_int, _vp]),
    "dsc_debug_set_gemm_stages": (None, [ctypes.c_int]),
    "dsc_linear_qkv_f16": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, ctypes.c_float, ctypes.c_int, _vp]),
    "dsc_add_layernorm": (ctypes.c_int, [_vp] * 6 + [ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp]),
    "dsc_geglu": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_linear_splitk_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dsc_linear_splitk_f16": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 3 +
                              [ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_int, _vp]),
    "dsc_conv3x3_gn_rows": (ctypes.c_int, [ctypes.c_int] * 7),
    "dsc_conv3x3_gn_nhwc_f16": (ctypes.c_int, [_vp] * 4 + [ctypes.c_int64, _vp, _vp] + [ctypes.c_int] * 5 + [ctypes.c_int64] * 3 +
                                [ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_linear_gn_rows": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dsc_linear_gn_f16": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 3 +
                          [ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_groupnorm_apply_nhwc": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp]),
    "dsc_softmax_rows_f16": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_float,
                                            ctypes.c_int, _vp]),
}


def load_library():
    """Load libdsc_hip.so once; raise DscLibraryError (never fall back) when it is absent or stale."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: its wheel carries its own libamdhip64 / libhipblaslt (same SONAMEs as /opt/rocm's).  Whichever copy is
    # loaded first serves the whole process; if this library pulled in /opt/rocm's before torch initialised, torch's
    # streams and this library's launches would sit on a runtime torch was not built against (observed: every launch
    # fails with hipErrorInvalidResourceHandle-class errors).  The streams handed to the C ABI are torch's.
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise DscLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP hot path.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # missing libamdhip64 etc.
        raise DscLibraryError(f"cannot load {path}: {e}") from e
    for name in declared_symbols():
        if not hasattr(lib, name):
            raise DscLibraryError(f"{path} does not export {name} (declared in include/dsc_hip.h): stale build?")
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if os.environ.get("DSC_LIBRARY_GEMM", "0") != "0" and not lib.dsc_has_library_gemm():
        raise DscLibraryError(f"DSC_LIBRARY_GEMM=1 asks for the hipBLASLt routing, but {path} was built without it: rebuild with "
                              "`DSC_WITH_HIPBLASLT=1 python -m diffusionspatialcontrol_amd.build --force`")
    if os.environ.get("DSC_SA_VARIANT"):                       # A/B switch: force one tiling of the flash self-attention kernel
        lib.dsc_debug_set_self_attn_variant(int(os.environ["DSC_SA_VARIANT"]))
    if os.environ.get("DSC_TUNING_PROFILE"):                   # latency (default) / throughput: dsc_set_tuning_profile at load time
        lib.dsc_set_tuning_profile({"latency": 0, "throughput": 1}[os.environ["DSC_TUNING_PROFILE"]])
    if os.environ.get("DSC_CONV_RING"):                        # A/B switch: weight-tile ring depth of the 3x3 convolution (3 / 9)
        lib.dsc_debug_set_conv_ring(int(os.environ["DSC_CONV_RING"]))
    if os.environ.get("DSC_GEMM_WIDE_MIN"):                    # A/B switch: workgroups a grid must keep for 128-column GEMM tiles
        lib.dsc_debug_set_gemm_stages(-int(os.environ["DSC_GEMM_WIDE_MIN"]))
    if os.environ.get("DSC_GN_MODE"):                          # A/B switch: GroupNorm kernel selection (dsc_debug_set_gn_mode)
        lib.dsc_debug_set_gn_mode(int(os.environ["DSC_GN_MODE"]))
    if os.environ.get("DSC_GEMM_STAGES"):                      # A/B switch: K-tile ring depth of the hand-written GEMM (2 / 3)
        lib.dsc_debug_set_gemm_stages(int(os.environ["DSC_GEMM_STAGES"]))
    _LIB = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load_library().dsc_status_string(status).decode()
        raise DscLibraryError(f"{what} failed with status {status}: {msg}")
