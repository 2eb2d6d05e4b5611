"""Thin torch-tensor wrappers over the C ABI (include/dsc_hip.h).  torch is plumbing here: device memory,
the current HIP stream and strides; all arithmetic happens in libdsc_hip.so."""
import ctypes

import torch

from . import _lib

FLAG_REF_FP16_ROUNDING = 1
FLAG_BIAS_IS_FINAL = 2

_WS = {}


def _stream_ptr(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _i64x3(a, b, c):
    return (ctypes.c_int64 * 3)(a, b, c)


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DscLibraryError("dsc ops run on the GPU only (no CPU fallback); got a CPU tensor")


def _workspace(device, nbytes):
    """A per-(device, stream) fp64 scratch buffer; grows monotonically, never shrinks (graph-capture safe
    once warmed up at the largest size)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 8 < nbytes:
        buf = torch.empty(max(nbytes, 1 << 16) // 8 + 1, dtype=torch.float64, device=device)
        _WS[key] = buf
    return buf


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
                 ref_fp16_rounding=True, bias_is_final=False, out=None):
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
    flags = (FLAG_REF_FP16_ROUNDING if ref_fp16_rounding else 0) | (FLAG_BIAS_IS_FINAL if bias_is_final else 0)
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


def region_xattn_std(q, k, *, layout="bhld", n_std_groups=1, scale=None, ref_fp16_rounding=True):
    """std of scale*q.k^T per std group (dsc_region_xattn_std) -> fp32 CUDA tensor [n_std_groups]."""
    _require_gpu(q, k)
    lib = _lib.load_library()
    lay = "bhld" if layout == "bhld" else "blc"
    qs, (Bc, H, L, d) = _blhd_strides(q, lay)
    ks, (_, _, S, _) = _blhd_strides(k, lay)
    out = torch.empty(n_std_groups, dtype=torch.float32, device=q.device)
    nbytes = lib.dsc_region_xattn_workspace_bytes(Bc, H, L, S, d, n_std_groups)
    ws = _workspace(q.device, nbytes)
    rc = lib.dsc_region_xattn_std(
        ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(k.data_ptr()), Bc, H, L, S, d, n_std_groups,
        _i64x3(*qs), _i64x3(*ks), float(scale) if scale else 0.0, 0,
        FLAG_REF_FP16_ROUNDING if ref_fp16_rounding else 0, ctypes.c_void_p(out.data_ptr()),
        ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _stream_ptr(q))
    _lib.check(rc, "dsc_region_xattn_std")
    return out
