"""Thin torch-tensor wrappers over the C ABI (include/dsc_hip.h).  torch is plumbing here: device memory,
the current HIP stream and strides; all arithmetic happens in libdsc_hip.so."""
import ctypes
import os

import numpy as np
import torch

from . import _lib

FLAG_REF_FP16_ROUNDING = 1
FLAG_BIAS_IS_FINAL = 2
FLAG_REUSE_STATS = 4
FLAG_ROWS_PADDED = 256
REGION_ROW_STRIDE = 100    # region_xattn_packed.hip kBP: the forward kernel's LDS bias-table row stride (floats)

def _stream_ptr(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _i64x3(a, b, c):
    return (ctypes.c_int64 * 3)(a, b, c)


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DscLibraryError("dsc ops run on the GPU only (no CPU fallback); got a CPU tensor")


def _workspace(device, nbytes):
    """fp64 scratch for one call, from torch's caching allocator.  Deliberately NOT cached across calls: inside a
    HIP-graph capture the allocation must belong to that graph's private pool, and a buffer remembered from an
    earlier (since destroyed) capture would alias live activations of the next one."""
    return torch.empty(max(nbytes, 16) // 8 + 1, dtype=torch.float64, device=device)


USE_CFG_SHARED_PREFIX = os.environ.get("DSC_CFG_PREFIX", "1") != "0"   # captured step: the layers in front of the first cross-attention once per image
USE_TEMB_HOIST = os.environ.get("DSC_TEMB_HOIST", "1") != "0"   # fused loop: the time-embedding path once per schedule, not per step


TUNING_PROFILES = {"latency": 0, "throughput": 1}       # DSC_TUNE_LATENCY / DSC_TUNE_THROUGHPUT (include/dsc_hip.h)


def set_tuning_profile(name):
    """launch rules for one generation at a time ("latency", the default) or for several generations in flight on their
    own streams ("throughput") - dsc_set_tuning_profile.  Read at launch time: a step graph keeps the profile it was
    captured under, and the pipeline re-captures a slot's step when the profile has changed since.  The package's own kernels
    give equal bytes under both; a GEMM left to the library (dsc_linear_lt_f16) may run another library algorithm and then
    agrees to rounding only."""
    if name not in TUNING_PROFILES:
        raise ValueError(f"tuning profile must be one of {sorted(TUNING_PROFILES)}")
    _lib.check(_lib.load_library().dsc_set_tuning_profile(TUNING_PROFILES[name]), "dsc_set_tuning_profile")


def tuning_profile():
    v = _lib.load_library().dsc_get_tuning_profile()
    return next(k for k, x in TUNING_PROFILES.items() if x == v)


def _blhd_strides(t, layout):
    """(sb, sl, sh) element strides of a [B, L, H, d]-addressable tensor.
    layout 'blc': t is [B, L, H*d] (or a view [B, L, H, d]);  'bhld': t is [B, H, L, d]."""
    if t.stride(-1) != 1:
        raise ValueError("innermost dimension must be contiguous")
    if layout == "bhld":
        B, H, L, d = t.shape
        return (t.stride(0), t.stride(2), t.stride(1)), (B, H, L, d)
    B, L, H, d = t.shape
    return (t.stride(0), t.stride(1), t.stride(2)), (B, H, L, d)


def region_xattn(q, k, v, region=None, sigma=1.0, *, layout="bhld", n_std_groups=1, scale=None,
                 ref_fp16_rounding=True, bias_is_final=False, out=None, reuse_stats=False, debug_flags=0):
    """softmax(scale*q.k^T + region*sigma*std) . v  on the GPU (dsc_region_xattn_fwd).

    layout 'bhld': q [Bc,H,L,d], k/v [Bc,H,S,d] -> out [Bc,H,L,d] (the shape of
                   scaled_dot_product_attention_regionstate, attention_modify.py:74);
    layout 'blhd': q [Bc,L,H,d], k/v [Bc,S,H,d] (views of the projection outputs) -> out [Bc,L,H,d] contiguous,
                   i.e. already the [Bc, L, H*d] tensor `to_out[0]` consumes.
    region: fp32 [Bw,L,S] on the same device or None.  sigma: python float or a 0-dim/1-element fp32 CUDA tensor.
    """
    _require_gpu(q, k, v, region)
    lib = _lib.load_library()
    if q.dtype != torch.float16 or k.dtype != torch.float16 or v.dtype != torch.float16:
        raise TypeError("region_xattn: fp16 tensors only (the dtype the reference pipeline runs in)")
    lay = "bhld" if layout == "bhld" else "blc"
    qs, (Bc, H, L, d) = _blhd_strides(q, lay)
    ks, (_, _, S, _) = _blhd_strides(k, lay)
    vs, _ = _blhd_strides(v, lay)
    if out is None:
        out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    os_, _ = _blhd_strides(out, lay)
    Bw = 0
    rptr = None
    if region is not None:
        if region.dtype != torch.float32:
            region = region.float()
        region = region.contiguous()
        if region.shape[1] != L or region.shape[2] != S:
            raise ValueError(f"region table {tuple(region.shape)} does not match L={L}, S={S}")
        Bw = region.shape[0]
        rptr = ctypes.c_void_p(region.data_ptr())
    sig_host, sig_dev = 0.0, None
    if isinstance(sigma, torch.Tensor):
        if sigma.is_cuda and sigma.dtype == torch.float32:
            sig_dev = ctypes.c_void_p(sigma.data_ptr())
        else:
            sig_host = float(sigma)
    else:
        sig_host = float(sigma)
    flags = (FLAG_REF_FP16_ROUNDING if ref_fp16_rounding else 0) | (FLAG_BIAS_IS_FINAL if bias_is_final else 0) \
        | (FLAG_REUSE_STATS if reuse_stats else 0) | debug_flags
    nbytes = lib.dsc_region_xattn_workspace_bytes(Bc, H, L, S, d, n_std_groups)
    ws = _workspace(q.device, nbytes)
    rc = lib.dsc_region_xattn_fwd(
        ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(v.data_ptr()),
        ctypes.c_void_p(out.data_ptr()), rptr, Bc, H, L, S, d, Bw, n_std_groups,
        _i64x3(*qs), _i64x3(*ks), _i64x3(*vs), _i64x3(*os_),
        sig_host, sig_dev, float(scale) if scale else 0.0, 0, flags,
        ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _stream_ptr(q))
    _lib.check(rc, "dsc_region_xattn_fwd")
    return out


def xattn_kv_pack(k, v, *, layout="blhd", out=None):
    """Pack cross-attention K / V ([Bc,S,H,d] views, or [Bc,H,S,d] with layout='bhld') into the MFMA-fragment image
    of dsc_xattn_kv_pack (once per generation: text keys/values are step-invariant)."""
    _require_gpu(k, v)
    lib = _lib.load_library()
    lay = "bhld" if layout == "bhld" else "blc"
    ks, (Bc, H, S, d) = _blhd_strides(k, lay)
    vs, _ = _blhd_strides(v, lay)
    nbytes = lib.dsc_xattn_kv_pack_bytes(Bc, H, S, d)
    if nbytes == 0:
        raise _lib.DscLibraryError(f"xattn_kv_pack: unsupported shape S={S}, d={d}")
    if out is None or out.numel() * 2 != nbytes:
        out = torch.empty(nbytes // 2, dtype=torch.float16, device=k.device)
    rc = lib.dsc_xattn_kv_pack(_p(k), _p(v), _p(out), Bc, H, S, d, _i64x3(*ks), _i64x3(*vs), 0, _stream_ptr(k))
    _lib.check(rc, "dsc_xattn_kv_pack")
    return out


MAX_REGION_ROWS = 32


def compress_region_table(w, pad_rows=False):
    """dense fp32 [Bw, L, S] -> (ids int16 [Bw, L], rows fp32 [NU, S]) with rows = the distinct table rows, or None
    when there are more than MAX_REGION_ROWS of them.  Lossless (rows[ids] == w).  Works on the tensor's own device:
    on a CPU table (where the reference keeps them, encode_region_map_function.py:33-34) nothing touches the GPU.
    pad_rows: zero-pad `rows` to MAX_REGION_ROWS so that shapes (and captured kernel arguments) never change."""
    Bw, L, S = w.shape
    flat = w.reshape(Bw * L, S)
    if not flat.is_cuda:
        # numpy (single-threaded, no torch intra-op pool): distinct rows through a 1-D key (two fixed random
        # projections in fp64), then verified exactly
        a = flat.contiguous().numpy()
        proj = np.random.default_rng(1234).random((S, 2))
        key = a.astype(np.float64) @ proj
        _, first_idx, inv_np = np.unique(key[:, 0] * 1.000123 + key[:, 1], return_index=True, return_inverse=True)
        if first_idx.shape[0] > MAX_REGION_ROWS:
            return None
        rows_np = a[first_idx]
        if not np.array_equal(rows_np[inv_np], a):       # projection collision (never seen): exact fallback
            rows_np, inv_np = np.unique(a, axis=0, return_inverse=True)
            if rows_np.shape[0] > MAX_REGION_ROWS:
                return None
        rows, inv = torch.from_numpy(np.ascontiguousarray(rows_np)), torch.from_numpy(inv_np.reshape(-1).astype(np.int64))
    else:
        rows, inv = torch.unique(flat, dim=0, return_inverse=True)
        if rows.shape[0] > MAX_REGION_ROWS:
            return None
    if pad_rows and rows.shape[0] < MAX_REGION_ROWS:
        rows = torch.cat([rows, rows.new_zeros(MAX_REGION_ROWS - rows.shape[0], S)])
    return inv.reshape(Bw, L).to(torch.int16).contiguous(), rows.contiguous()


def pad_region_rows(rows):
    """distinct region rows [NU, S] (compress_region_table) -> [NU, 100] fp32, zero padded: the shape of the forward kernel's LDS
    table, which it then fills with a flat 16-byte copy (DSC_FLAG_ROWS_PADDED).  S <= 96."""
    NU, S = rows.shape
    if S > 96:
        raise ValueError("pad_region_rows: at most 96 keys (one text chunk)")
    out = rows.new_zeros((NU, REGION_ROW_STRIDE), dtype=torch.float32)
    out[:, :S] = rows
    return out


def region_xattn_packed(q, packed_kv, S, region=None, sigma=1.0, *, n_std_groups=1, scale=None, ref_fp16_rounding=True,
                        out=None, reuse_stats=False, debug_flags=0):
    """dsc_region_xattn_fwd_packed: q [Bc,L,H,d] view, packed_kv from xattn_kv_pack, region = (ids, rows) from
    compress_region_table (device tensors; rows optionally through pad_region_rows) or None -> out [Bc,L,H,d] contiguous."""
    _require_gpu(q, packed_kv)
    lib = _lib.load_library()
    qs, (Bc, H, L, d) = _blhd_strides(q, "blc")
    if out is None:
        out = torch.empty((Bc, L, H, d), dtype=q.dtype, device=q.device)
    os_, _ = _blhd_strides(out, "blc")
    ids = rows = None
    Bw = nrows = 0
    if region is not None:
        ids, rows = region
        Bw, nrows = ids.shape[0], rows.shape[0]
    sig_host, sig_dev = 0.0, None
    if isinstance(sigma, torch.Tensor) and sigma.is_cuda and sigma.dtype == torch.float32:
        sig_dev = ctypes.c_void_p(sigma.data_ptr())
    else:
        sig_host = float(sigma)
    flags = (FLAG_REF_FP16_ROUNDING if ref_fp16_rounding else 0) | (FLAG_REUSE_STATS if reuse_stats else 0) | debug_flags
    if rows is not None and rows.shape[1] == REGION_ROW_STRIDE and S <= 96:      # pad_region_rows() form ([NU, S] has S <= 96 columns)
        flags |= FLAG_ROWS_PADDED
    ws = _workspace(q.device, lib.dsc_region_xattn_workspace_bytes(Bc, H, L, S, d, n_std_groups))
    rc = lib.dsc_region_xattn_fwd_packed(_p(q), _p(packed_kv), _p(out), _p(ids), _p(rows), nrows, Bc, H, L, S, d, Bw,
                                         n_std_groups, _i64x3(*qs), _i64x3(*os_), sig_host, sig_dev,
                                         float(scale) if scale else 0.0, 0, flags, _p(ws), ws.numel() * 8, _stream_ptr(q))
    _lib.check(rc, "dsc_region_xattn_fwd_packed")
    return out


def region_xattn_std(q, k, *, layout="bhld", n_std_groups=1, scale=None, ref_fp16_rounding=True, mask=None):
    """std of scale*q.k^T (+ mask) per std group (dsc_region_xattn_std[_masked]) -> fp32 CUDA tensor [n_std_groups].
    mask: additive fp32 CUDA tensor broadcastable to [Bc*H, L, S] (shapes [L, S], [1, S], [Bc*H, 1, S], [Bc*H, L, S], ...)."""
    _require_gpu(q, k, mask)
    lib = _lib.load_library()
    lay = "bhld" if layout == "bhld" else "blc"
    qs, (Bc, H, L, d) = _blhd_strides(q, lay)
    ks, (_, _, S, _) = _blhd_strides(k, lay)
    out = torch.empty(n_std_groups, dtype=torch.float32, device=q.device)
    nbytes = lib.dsc_region_xattn_workspace_bytes(Bc, H, L, S, d, n_std_groups)
    ws = _workspace(q.device, nbytes)
    flags = FLAG_REF_FP16_ROUNDING if ref_fp16_rounding else 0
    if mask is None:
        rc = lib.dsc_region_xattn_std(
            ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(k.data_ptr()), Bc, H, L, S, d, n_std_groups,
            _i64x3(*qs), _i64x3(*ks), float(scale) if scale else 0.0, 0, flags, ctypes.c_void_p(out.data_ptr()),
            ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _stream_ptr(q))
        _lib.check(rc, "dsc_region_xattn_std")
        return out
    m3 = mask.float()
    while m3.dim() < 3:
        m3 = m3.unsqueeze(0)
    if m3.dim() != 3 or m3.shape[2] != S or m3.shape[1] not in (1, L) or m3.shape[0] not in (1, Bc * H):
        raise ValueError(f"mask {tuple(mask.shape)} does not broadcast to [{Bc * H}, {L}, {S}]")
    m3 = m3.contiguous()
    ms = (ctypes.c_int64 * 2)(0 if m3.shape[0] == 1 else m3.stride(0), 0 if m3.shape[1] == 1 else m3.stride(1))
    rc = lib.dsc_region_xattn_std_masked(
        ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(k.data_ptr()), Bc, H, L, S, d, n_std_groups,
        _i64x3(*qs), _i64x3(*ks), float(scale) if scale else 0.0, 0, flags, ctypes.c_void_p(m3.data_ptr()), ms,
        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _stream_ptr(q))
    _lib.check(rc, "dsc_region_xattn_std_masked")
    return out


# ----------------------------------------------------------------------------- ops of the UNet step
def self_attention(q, k, v, scale=None, out=None):
    """softmax(q.k^T * scale) . v for q [B, L, H, d], k/v [B, S, H, d] (strided views allowed) -> [B, L, H, d]
    contiguous (dsc_self_attn_fwd: flash attention, scores never materialised)."""
    _drop_gn_partials(out)
    _require_gpu(q, k, v)
    if q.dtype != torch.float16:
        raise TypeError("self_attention: fp16 only")
    qs, (B, H, L, d) = _blhd_strides(q, "blc")
    ks, (_, _, S, _) = _blhd_strides(k, "blc")
    vs, _ = _blhd_strides(v, "blc")
    if out is None:
        out = torch.empty((B, L, H, d), dtype=q.dtype, device=q.device)
    os_, _ = _blhd_strides(out, "blc")
    rc = _lib.load_library().dsc_self_attn_fwd(
        ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(v.data_ptr()),
        ctypes.c_void_p(out.data_ptr()), B, H, L, S, d, _i64x3(*qs), _i64x3(*ks), _i64x3(*vs), _i64x3(*os_),
        float(scale) if scale else 0.0, 0, _stream_ptr(q))
    _lib.check(rc, "dsc_self_attn_fwd")
    return out


def groupnorm_silu(x, groups, weight, bias, eps, act):
    """GroupNorm(groups, eps) [+ SiLU] over NCHW fp16 [B, C, h, w] (dsc_groupnorm_silu: two HIP launches)."""
    _require_gpu(x)
    lib = _lib.load_library()
    if x.dtype != torch.float16:
        raise TypeError("groupnorm_silu: fp16 only")
    x = x.contiguous()
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // (B * C)
    y = torch.empty_like(x)
    ws = _workspace(x.device, lib.dsc_groupnorm_workspace_bytes(B, C, hw, groups))
    rc = lib.dsc_groupnorm_silu(_p(x), _p(y), _p(weight), _p(bias), B, C, hw, groups, float(eps), 1 if act else 0, 0,
                                _p(ws), ws.numel() * 8, _stream_ptr(x))
    _lib.check(rc, "dsc_groupnorm_silu")
    return y


def groupnorm_silu_nhwc(x, groups, weight, bias, eps, act, add=None):
    """GroupNorm(groups, eps) [+ SiLU] over channels-last fp16 (dsc_groupnorm_silu_nhwc).

    x: a channels_last [B, C, h, w] tensor (returned likewise) or token-major [B, hw, C].  add: optional [B, C] added
    to x first (the ResNet block's time-embedding term)."""
    _require_gpu(x)
    lib = _lib.load_library()
    if x.dtype != torch.float16:
        raise TypeError("groupnorm_silu_nhwc: fp16 only")
    if x.dim() == 4:
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        B, C, h, w = x.shape
        hw = h * w
        y = torch.empty_like(x, memory_format=torch.channels_last)
    else:
        x = x.contiguous()
        B, hw, C = x.shape
        y = torch.empty_like(x)
    add_stride = 0
    if add is not None:
        if add.stride(-1) != 1 or add.stride(0) % 8 != 0 or add.data_ptr() % 16 != 0:
            add = add.contiguous()
        add_stride = add.stride(0)
    ws = _workspace(x.device, lib.dsc_groupnorm_nhwc_workspace_bytes(B, C, hw, groups))
    rc = lib.dsc_groupnorm_silu_nhwc(_p(x), _p(y), _p(weight), _p(bias), _p(add), add_stride, B, C, hw, groups, float(eps),
                                     1 if act else 0, 0, _p(ws), ws.numel() * 8, _stream_ptr(x))
    _lib.check(rc, "dsc_groupnorm_silu_nhwc")
    return y


GN_CHECK = os.environ.get("DSC_GN_CHECK", "0") != "0"      # debug: verify a producer's partial sums against the tensor before use
USE_GN_FUSE = os.environ.get("DSC_GN_FUSE", "1") != "0"   # GroupNorm statistics from the producing convolution / GEMM epilogue (gn_partials.h)


class GnPartials:
    """the GroupNorm partial sums a producer emitted for the tensor it wrote (dsc_conv3x3_gn_nhwc_f16 / dsc_linear_gn_f16):
    buffer fp32 [B, rows, groups, 2, 2], for a tensor of C channels normalised in `groups` groups"""
    __slots__ = ("buf", "rows", "groups", "C", "B", "hw", "version")

    def __init__(self, buf, rows, groups, C, B, hw):
        self.buf, self.rows, self.groups, self.C, self.B, self.hw = buf, rows, groups, C, B, hw
        self.version = None                   # torch's in-place version counter of the tensor when the sums were attached


def attach_gn_partials(t, part):
    """hand a producer's partial sums on with the tensor OBJECT `t` (a view of the producer's output is fine: same bytes).

    CONTRACT: the sums are invalidated through torch's `_version` counter only, and this package's own kernels write through raw
    pointers (ctypes), which does not bump it - so NO dsc_* call may write in place into (or take as `out=`) a tensor that carries
    partials.  The ops of this module that accept `out=` or write in place drop the attribute from their destination
    (`_drop_gn_partials`); `DSC_GN_CHECK=1` makes `groupnorm_apply_nhwc` recompute the statistics from the tensor and compare."""
    if part is not None:
        part.version = t._version
        t._dsc_gn = part
    return t


def _drop_gn_partials(t):
    """a dsc_* kernel is about to write into `t` through its raw pointer: sums a producer attached to it no longer describe it"""
    if t is not None and getattr(t, "_dsc_gn", None) is not None:
        t._dsc_gn = None
    return t


def gn_partials_of(t):
    """the GnPartials a producer attached to the tensor OBJECT it returned (views and copies carry none), or None - also None once
    the tensor has been written in place since (torch's version counter): the sums would describe other bytes"""
    part = getattr(t, "_dsc_gn", None) if USE_GN_FUSE else None
    if part is not None and part.version != t._version:
        return None
    return part


def groupnorm_apply_nhwc(x, part, groups, weight, bias, eps, act):
    """GroupNorm [+ SiLU] in ONE launch from the producer's partial sums (dsc_groupnorm_apply_nhwc); x channels_last [B,C,h,w]"""
    _require_gpu(x)
    B, C, h, w = x.shape
    if part.groups != groups or part.C != C or part.B != B or part.hw != h * w or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("groupnorm_apply_nhwc: the partial sums do not belong to this tensor / grouping")
    if GN_CHECK:                               # debug: the producer's sums against statistics recomputed from the tensor itself
        cpg = C // groups
        pv = part.buf.view(B, part.rows, groups, 2, 2).double()
        straddles = torch.tensor([(g * cpg) // 64 != ((g + 1) * cpg - 1) // 64 for g in range(groups)], device=x.device)
        sums = pv[:, :, :, 0, 0].sum(1) + torch.where(straddles, pv[:, :, :, 1, 0].sum(1), torch.zeros((), dtype=pv.dtype, device=x.device))
        ref = x.double().reshape(B, groups, C // groups, h * w).sum(dim=(2, 3))
        if not torch.allclose(sums, ref, rtol=1e-3, atol=1e-2 * (C // groups) * h * w ** 0.5):
            raise RuntimeError("groupnorm_apply_nhwc: stale GroupNorm partial sums (the tensor was written after its producer)")
    y = torch.empty_like(x, memory_format=torch.channels_last)
    rc = _lib.load_library().dsc_groupnorm_apply_nhwc(_p(x), _p(y), _p(weight), _p(bias), _p(part.buf), part.rows, B, C, h * w,
                                                      groups, float(eps), 1 if act else 0, 0, _stream_ptr(x))
    _lib.check(rc, "dsc_groupnorm_apply_nhwc")
    return y


def conv3x3_gn_rows(x, weight, groups, upsample=False):
    """pixel tiles per image when dsc_conv3x3_gn_nhwc_f16 covers this convolution (and GroupNorm grouping of its output), else 0"""
    if not (USE_GN_FUSE and conv3x3_supported(x, weight, upsample=upsample)):
        return 0
    B, C, H, W = x.shape
    f = 2 if upsample else 1
    return int(_lib.load_library().dsc_conv3x3_gn_rows(B, H * f, W * f, C, weight.shape[0], groups, 1 if upsample else 0))


def conv3x3_gn(x, weight, groups, bias=None, add=None, residual=None, upsample=False):
    """conv3x3 (+ bias) (+ per-image bias row add [B, Cout]) (+ residual), channels_last; the returned tensor carries the
    GroupNorm partial sums of what was stored (gn_partials_of) - call only when conv3x3_gn_rows() > 0"""
    _require_gpu(x, weight)
    lib = _lib.load_library()
    cl = torch.channels_last
    if not x.is_contiguous(memory_format=cl):
        x = x.contiguous(memory_format=cl)
    if not weight.is_contiguous(memory_format=cl):
        weight = weight.contiguous(memory_format=cl)
    B, Cin, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    Cout = weight.shape[0]
    rows = int(lib.dsc_conv3x3_gn_rows(B, H, W, Cin, Cout, groups, 1 if upsample else 0))
    if rows <= 0:
        raise ValueError("conv3x3_gn: shape not covered (ask conv3x3_gn_rows first)")
    out = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=cl)
    ldr = 0
    if residual is not None:
        if residual.shape != out.shape:
            raise ValueError("conv3x3_gn: residual must have the output's shape")
        if not residual.is_contiguous(memory_format=cl):
            residual = residual.contiguous(memory_format=cl)
        ldr = Cout
    add_ld = 0
    if add is not None:
        if add.stride(-1) != 1 or add.stride(0) % 8 != 0 or add.data_ptr() % 16 != 0:
            add = add.contiguous()
        add_ld = add.stride(0)
    part = torch.empty((B, rows, groups, 2, 2), dtype=torch.float32, device=x.device)
    rc = lib.dsc_conv3x3_gn_nhwc_f16(_p(x), _p(weight), _p(bias), _p(add), add_ld, _p(residual), _p(out), B, H, W, Cin, Cout, Cin,
                                     ldr, Cout, 1 if upsample else 0, _p(part), groups, 0, _stream_ptr(x))
    _lib.check(rc, "dsc_conv3x3_gn_nhwc_f16")
    return attach_gn_partials(out, GnPartials(part, rows, groups, Cout, B, H * W))


def linear_gn(x, weight, bias, residual, rows_per_image, groups):
    """x [B, L, K] @ weight.T (+ bias) (+ residual) -> [B, L, N] with the GroupNorm partial sums of the result (a 1x1 convolution
    in token-major form: proj_out, conv_shortcut), or None when dsc_linear_gn_f16 does not cover the shape"""
    if not USE_GN_FUSE or not x.is_cuda or x.dtype != torch.float16 or x.dim() != 3:
        return None
    lib = _lib.load_library()
    B, L, K = x.shape
    N = weight.shape[0]
    M = B * L
    if L != rows_per_image or not weight.is_contiguous() or weight.dtype != torch.float16:
        return None
    rows = int(lib.dsc_linear_gn_rows(M, N, K, rows_per_image, groups))
    if rows <= 0 or not linear_kernel_can(x, weight, bias, residual):
        return None
    x2 = x.reshape(M, K)
    r2, ldr = _residual_2d(residual, M, N) if residual is not None else (None, 0)
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    part = torch.empty((B, rows, groups, 2, 2), dtype=torch.float32, device=x.device)
    rc = lib.dsc_linear_gn_f16(_p(x2), _p(weight), _p(bias), _p(r2), _p(out), M, N, K, x2.stride(0),
                               ldr, N, rows_per_image, _p(part), groups, 0, _stream_ptr(x))
    _lib.check(rc, "dsc_linear_gn_f16")
    return out.reshape(B, L, N), GnPartials(part, rows, groups, N, B, L)


USE_RESIDUAL_WRAP = os.environ.get("DSC_RESIDUAL_WRAP", "1") != "0"   # (A/B switch: 0 = repeat a shorter residual on the host, as before)


def _residual_2d(residual, M, N):
    """(r2, ldr) for the GEMM entry points: the residual as [R, N] rows and its row stride, with R << 32 in the stride's high
    bits when it has fewer rows than the result (include/dsc_hip.h, dsc_linear_f16: row m adds residual row m % R - a residual
    stream computed once per image under the shared CFG prefix); None when the shape is not one the kernels take."""
    R = residual.numel() // N
    if R * N != residual.numel() or R <= 0 or M % R != 0:
        return None
    r2 = residual.reshape(R, N)
    if r2.stride(1) != 1 or r2.stride(0) % 8 != 0 or r2.data_ptr() % 16 != 0:
        return None
    if R == M:
        return r2, r2.stride(0)
    if R % 128 != 0 or not USE_RESIDUAL_WRAP:
        return None
    return r2, r2.stride(0) | (R << 32)


def linear_kernel_can(x, weight, bias, residual):
    """dsc_linear_f16's operand constraints (not its row / K preferences): K, N multiples of 64, 16-byte aligned unit-stride rows"""
    N, K = weight.shape
    M = x.numel() // K
    if not (x.dtype == torch.float16 and weight.dtype == torch.float16 and K % 64 == 0 and N % 64 == 0 and x.stride(-1) == 1
            and weight.is_contiguous()):
        return False
    x2 = x.reshape(M, K)
    if not (x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0):
        return False
    if bias is not None and (bias.dtype != torch.float16 or bias.data_ptr() % 16 != 0):
        return False
    if residual is not None and _residual_2d(residual, M, N) is None:
        return False
    return True


USE_GN_CAT = os.environ.get("DSC_GN_CAT", "1") != "0"   # up blocks: the skip concatenation is written by the GroupNorm that reads it


def groupnorm_cat_covers(x1, x2):
    """True when dsc_groupnorm_silu_nhwc_cat takes this pair of channels_last [B, C1, h, w] / [B, C2, h, w] tensors"""
    cl = torch.channels_last
    return (USE_GN_CAT and x1.is_cuda and x1.dtype == torch.float16 and x2.dtype == torch.float16 and x1.dim() == 4
            and x2.dim() == 4 and x1.shape[0] == x2.shape[0] and x1.shape[2:] == x2.shape[2:] and x1.shape[1] % 8 == 0
            and x2.shape[1] % 8 == 0 and (x1.shape[1] + x2.shape[1]) // 8 <= 512
            and x1.is_contiguous(memory_format=cl) and x2.is_contiguous(memory_format=cl))


def groupnorm_silu_nhwc_cat(x1, x2, groups, weight, bias, eps, act, add=None):
    """(GroupNorm [+ SiLU] of cat([x1, x2], dim=1), the concatenation itself), both channels_last: the statistics pass reads
    the two sources and writes the concatenation on the way (dsc_groupnorm_silu_nhwc_cat) - no separate cat kernel."""
    _require_gpu(x1, x2)
    if not groupnorm_cat_covers(x1, x2):
        raise ValueError("groupnorm_silu_nhwc_cat: channels_last fp16 [B, C, h, w] pairs with C % 8 == 0 only")
    lib = _lib.load_library()
    B, C1, h, w = x1.shape
    C = C1 + x2.shape[1]
    cat = torch.empty((B, C, h, w), dtype=x1.dtype, device=x1.device, memory_format=torch.channels_last)
    y = torch.empty_like(cat, memory_format=torch.channels_last)
    add_stride = 0
    if add is not None:
        if add.stride(-1) != 1 or add.stride(0) % 8 != 0 or add.data_ptr() % 16 != 0:
            add = add.contiguous()
        add_stride = add.stride(0)
    ws = _workspace(x1.device, lib.dsc_groupnorm_nhwc_workspace_bytes(B, C, h * w, groups))
    rc = lib.dsc_groupnorm_silu_nhwc_cat(_p(x1), _p(x2), C1, _p(cat), _p(y), _p(weight), _p(bias), _p(add), add_stride, B, C,
                                         h * w, groups, float(eps), 1 if act else 0, 0, _p(ws), ws.numel() * 8,
                                         _stream_ptr(x1))
    _lib.check(rc, "dsc_groupnorm_silu_nhwc_cat")
    return y, cat


# Library GEMMs (hipBLASLt through dsc_linear_lt_f16) are OFF by default since round 3: every linear of the step runs on the
# package's own kernels (gemm_tn_f16, split-K for the few-row long-K shapes).  Per step that costs ~20 us against the best of
# both per shape (tools/mb_gemm_cold.py) and buys: the same bits in every process, on every rank and under both tuning profiles
# (the library's algorithm was timed per process and differed per profile), no 0.3 s of algorithm timing at start-up, and no
# stream-K kernels in a step that runs on two streams.  DSC_LIBRARY_GEMM=1 restores the round-2 routing for A/B measurements.
USE_LIBRARY_GEMM = os.environ.get("DSC_LIBRARY_GEMM", "0") != "0"
USE_LT_RESIDUAL = USE_LIBRARY_GEMM and os.environ.get("DSC_LT_RESIDUAL", "1") != "0"   # bias + residual in the library launch (dsc_linear_lt_f16)
USE_LT_ALL = USE_LIBRARY_GEMM and os.environ.get("DSC_LT_ALL", "1") != "0"   # ... and the ones without a residual go the same way
USE_DSC_GEMM = True        # route qualifying linears to dsc_linear_f16 (False: always hipBLASLt through torch)
DSC_GEMM_MIN_ROWS = 1024   # measured (tools/mb_gemm.py): the 128x64x64-tile kernel beats hipBLASLt + separate epilogue
DSC_GEMM_MAX_K = int(os.environ.get("DSC_GEMM_MAX_K", "640"))       # kernels for >= 1024 token rows and K <= 640; hipBLASLt's larger macro-tiles win beyond
USE_LN_FOLD = os.environ.get("DSC_LN_FOLD", "1") != "0"   # BasicTransformerBlock: LayerNorms folded into the GEMMs (dsc_linear_ln_f16)


DSC_GEMM_MID_ROWS = int(os.environ.get("DSC_GEMM_MID_ROWS", "128")) if os.environ.get("DSC_GEMM_MID", "1") != "0" else 1 << 30    # 128 <= rows < 1024 (the 16x16 and 8x8 levels at batch 1): the kernel up to K = 1280 - a wash per GEMM against the
DSC_GEMM_MID_K = int(os.environ.get("DSC_GEMM_MID_K", "1280"))      # library (QKV 14.0 vs 15.4 us, C->C 11.8 vs 11.1), but it lets the block's three LayerNorms fold into
#                            its GEMMs (three add+LayerNorm launches of 6.5 us fewer) and K / V leave the QKV GEMM head-major; the 8x8 level
#                            (128 rows) the same: equal time in the step and in images/s, three launches fewer (tools/ab_bench.sh)


def _gemm_rows_k_preferred(M, K, geglu=False):
    """row / K window in which the hand-written GEMM is the faster (or launch-saving) choice; tools/mb_gemm.py"""
    if M >= DSC_GEMM_MIN_ROWS and K <= DSC_GEMM_MAX_K:
        return True
    if DSC_GEMM_MID_ROWS <= M < DSC_GEMM_MIN_ROWS and K <= DSC_GEMM_MID_K:
        return True
    # GEGLU: the fused epilogue beats library GEMM + separate GEGLU launch at every row count of the UNet (M=512 N=10240
    # K=1280 22.2 vs 23.2 + the launch; M=128 12.5 vs 14.9)
    return bool(geglu) and K <= 1280


def linear_kernel_covers(M, N, K, dtype, geglu=False):
    """True when dsc_linear_f16 (the hand-written MFMA GEMM with fused epilogues) takes this shape"""
    return (USE_DSC_GEMM and dtype == torch.float16 and K % 64 == 0 and N % 64 == 0 and _gemm_rows_k_preferred(M, K, geglu)
            and (not geglu or (N // 2) % 32 == 0))


def fold_layernorm(weight, bias, gamma, beta):
    """(w', b', cvec) for dsc_linear_ln_f16: LayerNorm(s; gamma, beta) @ weight.T + bias
       == rstd (s @ w'.T - mu cvec) + b'   with w' = weight * gamma, b' = weight @ beta + bias, cvec = w'.sum(1) (fp32, of
    the fp16-rounded w' the kernel multiplies with)."""
    w2 = (weight.float() * gamma.float()[None, :]).to(weight.dtype).contiguous()
    b2 = weight.float() @ beta.float()
    if bias is not None:
        b2 = b2 + bias.float()
    return w2, b2.to(weight.dtype).contiguous(), w2.float().sum(dim=1).contiguous()


def linear_ln(x, weight, bias, *, residual=None, geglu=False, ln=None, ln_stats=False):
    """dsc_linear_ln_f16 (no library fallback: call only when linear_kernel_covers() says so).
    ln = (partials [M, nb, 2] fp32, cvec [N] fp32, eps): x is the un-normalised stream, weight / bias come from
    fold_layernorm().  ln_stats=True additionally returns the [M, N/64, 2] row partials of the output."""
    _require_gpu(x, weight)
    N, K = weight.shape
    lead = x.shape[:-1]
    M = 1
    for v in lead:
        M *= v
    x2 = x.reshape(M, K)
    if x2.stride(1) != 1 or x2.stride(0) % 8 != 0 or not weight.is_contiguous():
        raise ValueError("linear_ln: unit inner stride, 16-byte aligned rows and a contiguous weight are required")
    r2, ldr = None, 0
    if residual is not None:
        got = _residual_2d(residual, M, N)
        if got is None:
            raise ValueError("linear_ln: residual rows must be 16-byte aligned with unit inner stride (M rows, or M / k rows in multiples of 128)")
        r2, ldr = got
    n_out = N // 2 if geglu else N
    out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    stats = torch.empty((M, N // 64, 2), dtype=torch.float32, device=x.device) if ln_stats else None
    part, cvec, eps, nb = None, None, 0.0, 0
    if ln is not None:
        part, cvec, eps = ln
        nb = part.shape[1]
        if part.shape[0] != M or part.dtype != torch.float32 or not part.is_contiguous() or cvec.numel() != N:
            raise ValueError("linear_ln: statistics / cvec do not match the operands")
    rc = _lib.load_library().dsc_linear_ln_f16(_p(x2), _p(weight), _p(bias), _p(r2), _p(out), M, N, K, x2.stride(0),
                                               ldr, n_out, 1 if geglu else 0,
                                               _p(part), nb, _p(cvec), float(eps), _p(stats), 0, _stream_ptr(x))
    _lib.check(rc, "dsc_linear_ln_f16")
    out = out.reshape(*lead, n_out)
    return (out, stats) if ln_stats else out


USE_QKV_HEAD_MAJOR = os.environ.get("DSC_QKV_HEAD_MAJOR", "1") != "0"


def linear_qkv_covers(x, weight, heads):
    """True when dsc_linear_qkv_f16 takes the fused self-attention projection of x [B, L, K] by weight [3C, K]"""
    if not (USE_QKV_HEAD_MAJOR and x.dim() == 3 and x.dtype == torch.float16 and weight.dtype == torch.float16):
        return False
    B, L, K = x.shape
    C = weight.shape[0] // 3
    return (weight.shape[0] == 3 * C and C % 64 == 0 and C % heads == 0 and (C // heads) % 8 == 0
            and linear_kernel_covers(B * L, 3 * C, K, x.dtype) and x.stride(-1) == 1 and weight.is_contiguous())


def linear_qkv(x, weight, bias, heads, ln=None):
    """The self-attention q / k / v projection as one GEMM with the head split of K and V done by its epilogue
    (dsc_linear_qkv_f16): x [B, L, K], weight [3C, K] -> (q4, k4, v4), each a [B, L, heads, d] VIEW: q4 of a [B, L, C]
    tensor, k4 / v4 of a head-major [2, B, heads, L, d] one (a head's keys / values contiguous).  ln as in linear_ln."""
    _require_gpu(x, weight)
    B, L, K = x.shape
    C = weight.shape[0] // 3
    d = C // heads
    M = B * L
    x2 = x.reshape(M, K)
    if x2.stride(1) != 1 or x2.stride(0) % 8 != 0:
        raise ValueError("linear_qkv: unit inner stride and 16-byte aligned rows are required")
    q = torch.empty((B, L, C), dtype=x.dtype, device=x.device)
    kv = torch.empty((2, B, heads, L, d), dtype=x.dtype, device=x.device)
    part, cvec, eps, nb = None, None, 0.0, 0
    if ln is not None:
        part, cvec, eps = ln
        nb = part.shape[1]
        if part.shape[0] != M or part.dtype != torch.float32 or not part.is_contiguous() or cvec.numel() != 3 * C:
            raise ValueError("linear_qkv: statistics / cvec do not match the operands")
    rc = _lib.load_library().dsc_linear_qkv_f16(_p(x2), _p(weight), _p(bias), _p(q), _p(kv), M, C, K, x2.stride(0), C, heads, L,
                                                _p(part), nb, _p(cvec), float(eps), 0, _stream_ptr(x))
    _lib.check(rc, "dsc_linear_qkv_f16")
    return q.view(B, L, heads, d), kv[0].permute(0, 2, 1, 3), kv[1].permute(0, 2, 1, 3)


USE_SPLITK = os.environ.get("DSC_GEMM_SPLITK", "1") != "0"    # few-row long-K GEMMs on dsc_linear_splitk_f16 instead of the library
SPLITK_MAX_ROWS = int(os.environ.get("DSC_SPLITK_MAX_ROWS", "512"))
SPLITK_MIN_K = int(os.environ.get("DSC_SPLITK_MIN_K", "1920"))


def linear_splitk(x2, weight, bias, r2, splits=0):
    """dsc_linear_splitk_f16 on 2-D operands (x2 [M, K], weight [N, K] contiguous, r2 [M, N] or None) -> [M, N]"""
    lib = _lib.load_library()
    M, K = x2.shape
    N = weight.shape[0]
    out = torch.empty((M, N), dtype=x2.dtype, device=x2.device)
    nbytes = lib.dsc_linear_splitk_workspace_bytes(M, N, K, splits)
    ws = _workspace(x2.device, nbytes) if nbytes else None
    rc = lib.dsc_linear_splitk_f16(_p(x2), _p(weight), _p(bias), _p(r2), _p(out), M, N, K, x2.stride(0),
                                   r2.stride(0) if r2 is not None else 0, N, splits, _p(ws), ws.numel() * 8 if ws is not None else 0,
                                   0, _stream_ptr(x2))
    _lib.check(rc, "dsc_linear_splitk_f16")
    return out


def linear(x, weight, bias=None, residual=None, geglu=False, prefer_kernel=False):
    """x @ weight.T (+ bias) (+ residual), or the fused GEGLU of [x @ weight.T + bias]; x [..., K], weight [N, K].
    prefer_kernel: take the hand-written GEMM whenever it CAN run the shape, whatever the row / K thresholds say.

    Shapes the hand-written MFMA kernel covers (fp16, K % 64 == 0, N % 64 == 0, >= DSC_GEMM_MIN_ROWS rows, unit inner
    stride) go to dsc_linear_f16 with the epilogue fused; everything else is a plain library GEMM through torch
    (hipBLASLt) followed by the separate epilogue ops."""
    _require_gpu(x, weight)
    N, K = weight.shape
    lead = x.shape[:-1]
    M = 1
    for v in lead:
        M *= v
    # `can`: dsc_linear_f16 is ABLE to run the shape; `ok`: it is also the faster choice (row / K thresholds)
    can = (USE_DSC_GEMM and x.dtype == torch.float16 and weight.dtype == torch.float16 and K % 64 == 0 and N % 64 == 0
           and x.stride(-1) == 1 and weight.is_contiguous()
           and (not geglu or (bias is not None and residual is None and (N // 2) % 32 == 0)))
    x2 = None
    if can:
        x2 = x.reshape(M, K)                      # a view when the leading dims collapse (the token-major case)
        can = x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0
    ldr = 0
    if residual is not None and residual.numel() != M * N:
        # fewer residual rows than result rows (shared CFG prefix): the kernel wraps them (dsc_linear_f16), every other route
        # gets them repeated
        got = _residual_2d(residual, M, N) if can and not geglu else None
        if got is None or not (prefer_kernel or _gemm_rows_k_preferred(M, K, geglu)):
            residual = residual.reshape(-1, N).repeat(M // (residual.numel() // N), 1).reshape(*lead, N)
    if can and residual is not None:
        got = _residual_2d(residual, M, N)
        can = got is not None
        if can:
            r2, ldr = got
    ok = can and (prefer_kernel or _gemm_rows_k_preferred(M, K, geglu))
    if can and not ok and not geglu and USE_SPLITK and not USE_LIBRARY_GEMM and M <= SPLITK_MAX_ROWS and K >= SPLITK_MIN_K \
            and (bias is None or (bias.dtype == torch.float16 and bias.data_ptr() % 16 == 0)):
        # few rows, long K (the 16x16 / 8x8 levels' feed-forward output projections and shortcut 1x1s): weight-streaming bound,
        # split-K over workgroups + an ordered reduce launch (bias / residual there) - measured against the library per shape
        # in the step's cache state (tools/mb_gemm_cold.py)
        return linear_splitk(x2, weight, bias, residual.reshape(M, N) if residual is not None else None).reshape(*lead, N)
    if not ok:
        if ((USE_LT_RESIDUAL if residual is not None else USE_LT_ALL) and x.dtype == torch.float16 and K % 8 == 0
                and N % 8 == 0 and M >= 8 and not (geglu and residual is not None)):
            # library GEMM with the bias epilogue AND the residual as beta*C: one launch instead of GEMM + add; the
            # algorithm is the fastest of the heuristic's candidates, timed on the first call of a shape
            xl = x.reshape(M, K)
            rl = residual.reshape(M, N) if residual is not None else None
            if (xl.stride(1) == 1 and xl.stride(0) % 8 == 0 and weight.is_contiguous() and weight.dtype == torch.float16
                    and (bias is None or bias.dtype == torch.float16)
                    and (rl is None or (rl.stride(1) == 1 and rl.stride(0) % 8 == 0))):
                out = torch.empty((M, N), dtype=x.dtype, device=x.device)
                rc = _lib.load_library().dsc_linear_lt_f16(_p(xl), _p(weight), _p(bias), _p(rl), _p(out), M, N, K,
                                                           xl.stride(0), rl.stride(0) if rl is not None else 0, N, 0,
                                                           _stream_ptr(x))
                if rc == 0:
                    out = out.reshape(*lead, N)
                    return globals()["geglu"](out) if geglu else out
        # dsc_linear_lt_f16 declined (no workspace-free library algorithm for the shape, linear_lt.hip): the hand-written
        # kernel takes it when it can, so that no GEMM of the step reaches a library algorithm this package did not vet
        ok = can
    if not ok:
        y = torch.nn.functional.linear(x, weight, bias)
        if geglu:
            return globals()["geglu"](y)
        return y if residual is None else y + residual
    n_out = N // 2 if geglu else N
    out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    rc = _lib.load_library().dsc_linear_f16(_p(x2), _p(weight), _p(bias), _p(r2) if residual is not None else None, _p(out),
                                            M, N, K, x2.stride(0), ldr if residual is not None else 0, n_out,
                                            1 if geglu else 0, 0, _stream_ptr(x))
    _lib.check(rc, "dsc_linear_f16")
    return out.reshape(*lead, n_out)


USE_DSC_CONV = True        # route qualifying 3x3 convolutions to dsc_conv3x3_nhwc_f16 (False: always MIOpen through torch)


def conv3x3_supported(x, weight, upsample=False):
    """True when dsc_conv3x3_nhwc_f16 covers this [B, Cin, H, W] channels_last fp16 input / [Cout, Cin, 3, 3] weight
    (upsample: the convolution runs on the 2x nearest-upsampled image)."""
    if not (USE_DSC_CONV and x.is_cuda and x.dtype == torch.float16 and weight.dtype == torch.float16 and x.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] == x.shape[1]):
        return False
    B, C, H, W = x.shape
    f = 2 if upsample else 1
    return bool(_lib.load_library().dsc_conv3x3_supported(B, H * f, W * f, C, weight.shape[0]))


def conv3x3(x, weight, bias=None, residual=None, splits=0, upsample=False, out_nchw=False, stride2=False, stride2_pad_br=False):
    """3x3 / stride 1 / pad 1 convolution (+ bias) (+ residual) of a channels_last fp16 [B, Cin, H, W] tensor with a
    [Cout, Cin, 3, 3] weight held in channels_last memory format (dsc_conv3x3_nhwc_f16); returns channels_last
    [B, Cout, H, W], or a plain contiguous (NCHW) tensor with out_nchw=True.  upsample=True convolves the 2x
    nearest-neighbour upsampling of x (output [B, Cout, 2H, 2W]) without materialising it; stride2=True is the
    stride-2 / pad-1 convolution (output [B, Cout, H/2, W/2], H and W even); stride2_pad_br=True the stride-2 convolution with
    zero padding on the bottom / right only (`F.pad(x, (0, 1, 0, 1))` + stride-2 / pad-0 conv: AutoencoderKL encoder).  Raises
    on an unsupported shape - ask conv3x3_supported() first."""
    _require_gpu(x, weight)
    lib = _lib.load_library()
    cl = torch.channels_last
    if not x.is_contiguous(memory_format=cl):
        x = x.contiguous(memory_format=cl)
    if not weight.is_contiguous(memory_format=cl):
        weight = weight.contiguous(memory_format=cl)
    B, Cin, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    Cout = weight.shape[0]
    if int(bool(upsample)) + int(bool(stride2)) + int(bool(stride2_pad_br)) > 1:
        raise ValueError("conv3x3: upsample, stride2 and stride2_pad_br exclude each other")
    oh, ow = (H // 2, W // 2) if (stride2 or stride2_pad_br) else (H, W)
    if out_nchw:
        out = torch.empty((B, Cout, oh, ow), dtype=x.dtype, device=x.device)
    else:
        out = torch.empty((B, Cout, oh, ow), dtype=x.dtype, device=x.device, memory_format=cl)
    ldr = 0
    if residual is not None:
        if residual.shape != out.shape:
            raise ValueError("conv3x3: residual must have the output's shape")
        if not residual.is_contiguous(memory_format=cl):
            residual = residual.contiguous(memory_format=cl)
        ldr = Cout
    nbytes = lib.dsc_conv3x3_workspace_bytes(B, H, W, Cin, Cout, splits)
    ws = _workspace(x.device, nbytes) if nbytes else None
    rc = lib.dsc_conv3x3_nhwc_f16(_p(x), _p(weight), _p(bias), _p(residual), _p(out), B, H, W, Cin, Cout, Cin, ldr, Cout,
                                  3 if stride2_pad_br else (2 if stride2 else (1 if upsample else 0)), 1 if out_nchw else 0, splits, 0, _p(ws),
                                  ws.numel() * 8 if ws is not None else 0, _stream_ptr(x))
    _lib.check(rc, "dsc_conv3x3_nhwc_f16")
    return out


def conv3x3_fewcin(x, weight_t, bias, cout):
    """3x3 / pad 1 convolution of a plain (NCHW) fp16 [B, Cin <= 16, H, W] tensor -> channels_last [B, cout, H, W]
    (dsc_conv3x3_fewcin_f16); weight_t is weight.reshape(cout, Cin * 9).t().contiguous()."""
    _require_gpu(x, weight_t)
    x = x.contiguous()
    B, Cin, H, W = x.shape
    out = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rc = _lib.load_library().dsc_conv3x3_fewcin_f16(_p(x), _p(weight_t), _p(bias), _p(out), B, Cin, H, W, cout, 0, _stream_ptr(x))
    _lib.check(rc, "dsc_conv3x3_fewcin_f16")
    return out


def linear_rows(x, weight, bias=None, silu_out=False, sinusoid_dim=0):
    """Few-row linear (<= 8 rows) of the time-embedding path (dsc_linear_rows_f16): act(x @ weight.T + bias).
    sinusoid_dim > 0: x is an fp32 [M] tensor of timesteps and the input row is its sinusoidal embedding of that width
    ([cos | sin], diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0)), generated inside the kernel."""
    _require_gpu(x, weight)
    N, K = weight.shape
    if sinusoid_dim:
        if x.dtype != torch.float32 or x.dim() != 1 or sinusoid_dim != K:
            raise ValueError("linear_rows: sinusoid input is an fp32 [M] tensor and sinusoid_dim == weight.shape[1]")
        x = x.contiguous()                                   # (an expanded scalar timestep has stride 0: the kernel reads x[m])
        M, ldx = x.shape[0], 0
    else:
        x = x if x.stride(-1) == 1 else x.contiguous()
        M, ldx = x.shape[0], x.stride(0)
    out = torch.empty((M, N), dtype=weight.dtype, device=weight.device)
    flags = (1 if sinusoid_dim else 0) | (2 if silu_out else 0)
    rc = _lib.load_library().dsc_linear_rows_f16(_p(x), _p(weight), _p(bias), _p(out), M, N, K, ldx, N, flags, 0, _stream_ptr(weight))
    _lib.check(rc, "dsc_linear_rows_f16")
    return out


def add_bias_residual(a, b, bias=None):
    """a + b + bias[c] over channels-last / token-major fp16 tensors of identical layout (dsc_add_bias_residual)."""
    _require_gpu(a, b)
    if a.dim() == 4:
        cl = torch.channels_last
        if not a.is_contiguous(memory_format=cl):
            a = a.contiguous(memory_format=cl)
        if not b.is_contiguous(memory_format=cl):
            b = b.contiguous(memory_format=cl)
        C = a.shape[1]
        out = torch.empty_like(a, memory_format=cl)
    else:
        a, b = a.contiguous(), b.contiguous()
        C = a.shape[-1]
        out = torch.empty_like(a)
    rc = _lib.load_library().dsc_add_bias_residual(_p(a), _p(b), _p(bias), _p(out), a.numel() // C, C, 0, _stream_ptr(a))
    _lib.check(rc, "dsc_add_bias_residual")
    return out


def add_layernorm(x, a, weight, bias, eps=1e-5):
    """(x + a, LayerNorm(x + a)) in one launch (dsc_add_layernorm); a may be None -> (x, LayerNorm(x)).
    x, a: [..., C] fp16 contiguous."""
    _require_gpu(x)
    x = x.contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    s = x
    if a is not None:
        a = a.contiguous()
        s = torch.empty_like(x)
    rc = _lib.load_library().dsc_add_layernorm(_p(x), _p(a), _p(weight), _p(bias), _p(s) if a is not None else None, _p(y),
                                               rows, C, float(eps), 0, _stream_ptr(x))
    _lib.check(rc, "dsc_add_layernorm")
    return s, y


def softmax_rows(scores, scale=1.0, out=None):
    """softmax(scale * scores, dim=-1) of an fp16 [rows, n] matrix (unit inner stride, 16-byte aligned rows) in fp32 arithmetic
    (dsc_softmax_rows_f16) - the middle step of the VAE's 512-channel attention head between its two GEMMs."""
    _drop_gn_partials(out)
    _require_gpu(scores)
    if scores.dtype != torch.float16 or scores.dim() != 2 or scores.stride(1) != 1:
        raise TypeError("softmax_rows: a 2-D fp16 matrix with unit inner stride")
    if out is None:
        out = torch.empty_like(scores)
    rc = _lib.load_library().dsc_softmax_rows_f16(_p(scores), _p(out), scores.shape[0], scores.shape[1], scores.stride(0),
                                                  out.stride(0), float(scale), 0, _stream_ptr(scores))
    _lib.check(rc, "dsc_softmax_rows_f16")
    return out


def geglu(x):
    """hidden * gelu(gate) for x = [..., 2n] fp16 contiguous (dsc_geglu)."""
    _require_gpu(x)
    lib = _lib.load_library()
    x = x.contiguous()
    n = x.shape[-1] // 2
    y = torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)
    rc = lib.dsc_geglu(_p(x), _p(y), x.numel() // (2 * n), n, 0, _stream_ptr(x))
    _lib.check(rc, "dsc_geglu")
    return y


# ----------------------------------------------------------------------------- sampler step
GRAPHS_ENABLED = True      # the fused pipeline captures the UNet step into a HIP graph (torch.cuda.CUDAGraph)
PROTOCOL_GRAPH = os.environ.get("DSC_PROTOCOL_GRAPH", "1") != "0"   # protocol-mode model calls replay the same graph


def _row_args(row):
    """(src, dst, halfs, copies) of the optional row broadcast of the sampler kernels: row = (src [n] fp16, dst [copies, n] fp16)"""
    if row is None:
        return None, None, 0, 0
    src, dst = row
    _require_gpu(src, dst)
    if src.dtype != torch.float16 or dst.dtype != torch.float16 or not dst.is_contiguous() or src.stride(-1) != 1 \
            or dst.dim() != 2 or dst.shape[1] != src.numel():
        raise ValueError("row broadcast: fp16 source [n] and contiguous destination [copies, n]")
    return _p(src), _p(dst), src.numel(), dst.shape[0]


def prepare_unet_input(x, c_in, t, sigma, x_in, t_buf, sigma_buf, row=None):
    """x_in = [x; x] * c_in, t_buf[:] = t, sigma_buf[0] = sigma (dsc_prepare_unet_input); row: see _row_args."""
    _require_gpu(x, x_in, t_buf, sigma_buf)
    n_img = x.shape[0]
    rc = _lib.load_library().dsc_prepare_unet_input(_p(x), c_in, t, sigma, _p(x_in), _p(t_buf), _p(sigma_buf), n_img,
                                                    x.numel() // n_img, 0, *_row_args(row), _stream_ptr(x))
    _lib.check(rc, "dsc_prepare_unet_input")


def cfg_dpmpp2m_step(x, eps, old, sigma, guidance, a, b, c, c_in_next, t_next, sigma_next, x_in, t_buf, sigma_buf, row=None):
    """One launch: CFG combine + eps->denoised + DPM++ 2M update (in place on x, old) + next UNet input (+ the row broadcast)."""
    _require_gpu(x, eps, old, x_in)
    n_img = x.shape[0]
    rc = _lib.load_library().dsc_cfg_dpmpp2m_step(_p(x), _p(eps), _p(old), sigma, guidance, a, b, c, c_in_next, t_next,
                                                  sigma_next, _p(x_in), _p(t_buf), _p(sigma_buf), n_img,
                                                  x.numel() // n_img, 0, *_row_args(row), _stream_ptr(x))
    _lib.check(rc, "dsc_cfg_dpmpp2m_step")


def dpmpp2m_update(x, denoised, old, a, b, c):
    """a*x + b*denoised + c*old as one launch (dsc_dpmpp2m_update)."""
    _require_gpu(x, denoised)
    x, denoised = x.contiguous(), denoised.contiguous()
    if old is not None:
        old = old.contiguous()
    elif c != 0.0:
        raise ValueError("c != 0 needs the previous denoised estimate")
    out = torch.empty_like(x)
    rc = _lib.load_library().dsc_dpmpp2m_update(_p(x), _p(denoised), _p(old), a, b, c if old is not None else 0.0,
                                                _p(out), x.numel(), 0, _stream_ptr(x))
    _lib.check(rc, "dsc_dpmpp2m_update")
    return out
