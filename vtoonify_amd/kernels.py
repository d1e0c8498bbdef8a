"""Thin torch<->C-ABI adapters: one Python function per entry point of
include/vtoonify_amd.h.  Tensors are only used as device-memory handles (data_ptr) and to
pick the current HIP stream; all arithmetic happens in libvtoonify_amd.so.

GPU tensors are mandatory with the product library.  CPU tensors are accepted only when a
test has injected the host-emulation build (see _lib.use_library).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU_TANH, OUT_NCHW, OUT_NHWC, VT_BF16, VT_F16, VT_F32, VT_F32X3, VT_F64, ConvDesc

_DT = {torch.float32: VT_F32, torch.bfloat16: VT_BF16, torch.float16: VT_F16}


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise _lib.VtError(f"unsupported dtype {dtype}") from None


def op_dt_code(dtype: torch.dtype) -> int:
    """dtype code of the two native operators, which also take double like the reference's dispatch
    (upfirdn2d_kernel.cu:311, fused_bias_act_kernel.cu:96: AT_DISPATCH_FLOATING_TYPES_AND_HALF)."""
    return VT_F64 if dtype == torch.float64 else dt_code(dtype)


def _dev_ok(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda and not _lib.is_emulation():
            raise _lib.VtError(
                "vtoonify_amd kernels run on the GPU only (got a CPU tensor); the package has no CPU path")
        if not t.is_contiguous():
            raise _lib.VtError("tensor must be contiguous")


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


# ---------------------------------------------------------------------------------------
def upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    oh, ow = C.c_int(), C.c_int()
    _lib.check(_lib.lib().vt_upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1,
                                               py0, py1, C.byref(oh), C.byref(ow)), "vt_upfirdn2d_out_size")
    return oh.value, ow.value


def upfirdn2d_planes(x: torch.Tensor, fir: torch.Tensor, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """x: (planes, in_h, in_w) contiguous; fir: (kh, kw) fp32 (fp64 for an fp64 x) on the same device."""
    _dev_ok(x, fir)
    if fir.dtype != (torch.float64 if x.dtype == torch.float64 else torch.float32):
        raise _lib.VtError("upfirdn2d: FIR taps must be fp32 (fp64 for fp64 input)")
    planes, in_h, in_w = x.shape
    kh, kw = fir.shape
    oh, ow = upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
    if oh <= 0 or ow <= 0:
        raise _lib.VtError(f"upfirdn2d: empty output ({oh} x {ow})")
    out = torch.empty((planes, oh, ow), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().vt_upfirdn2d(_p(out), _p(x), _p(fir), planes, in_h, in_w, kh, kw, up_x, up_y,
                                      down_x, down_y, px0, px1, py0, py1, op_dt_code(x.dtype), _stream(x)),
               "vt_upfirdn2d")
    return out


def fused_bias_act(x, bias, refer, act, grad, alpha, scale):
    _dev_ok(x, bias, refer)
    out = torch.empty_like(x)
    step_b = 1
    for s in x.shape[2:]:
        step_b *= int(s)
    size_b = int(bias.numel()) if bias is not None else 1
    _lib.check(_lib.lib().vt_fused_bias_act(_p(out), _p(x), _p(bias), _p(refer), x.numel(), step_b, size_b,
                                           act, grad, float(alpha), float(scale), op_dt_code(x.dtype),
                                           _stream(x)), "vt_fused_bias_act")
    return out


def _ptr(v):
    if v is None:
        return None
    if isinstance(v, torch.Tensor):
        return v.data_ptr()
    return int(v)


def make_conv_desc(*, src0, c0, ld0, n, h, w, out_h, out_w, weight, cout, kh, kw, out, ld_out, dtype,
                   src1=None, c1=0, ld1=0, stride=1, pad=0, dil=1, phases=1, transposed=0, in_scale=None,
                   in_shift=None, bias=None, act=ACT_NONE, slope=0.2, gain=1.0, alpha=1.0, beta=0.0,
                   alpha_dev=None, resid=None, ld_res=0, out_layout=OUT_NHWC, out_dtype=None,
                   tile_hint=0, splitk_ws=None, slope_vec=None, rgb_weight=None, rgb_bias=None, rgb_resid=None,
                   rgb_out=None, stats_part=None, post_relu=0, weight_stream=None, tile_stats=None,
                   in_tile_stats=None, in_stats_dil=1, in_gb=None, in_ld_gb=0, up_fir=None, pad_w=None, rgb_only=0, in_absdiff=0) -> ConvDesc:
    """Fill a vt_conv_desc.  Pointers may be tensors or raw ints (sub-views: data_ptr()+offset)."""
    d = ConvDesc()
    d.src0, d.src1 = _ptr(src0), _ptr(src1)
    d.c0, d.c1, d.ld0, d.ld1 = c0, c1, ld0, ld1
    d.n, d.h, d.w, d.out_h, d.out_w = n, h, w, out_h, out_w
    d.weight = _ptr(weight)
    d.cout = cout
    d.kh, d.kw, d.stride, d.pad, d.dil = kh, kw, stride, pad, dil
    d.phases, d.transposed = phases, transposed
    d.in_scale, d.in_shift = _ptr(in_scale), _ptr(in_shift)
    d.bias = _ptr(bias)
    d.act, d.slope, d.gain, d.alpha, d.beta = act, slope, gain, alpha, beta
    d.alpha_dev = _ptr(alpha_dev)
    d.resid, d.ld_res = _ptr(resid), ld_res
    d.out, d.ld_out = _ptr(out), ld_out
    d.out_layout = out_layout
    d.dtype = dtype
    d.out_dtype = (VT_F32 if dtype == VT_F32X3 else dtype) if out_dtype is None else out_dtype   # (f32x3: fp32 tensors)
    d.tile_hint = tile_hint
    d.slope_vec = _ptr(slope_vec)
    d.rgb_weight, d.rgb_bias, d.rgb_resid, d.rgb_out = _ptr(rgb_weight), _ptr(rgb_bias), _ptr(rgb_resid), _ptr(rgb_out)
    d.stats_part = _ptr(stats_part)
    d.post_relu = int(post_relu)
    d.weight_stream = _ptr(weight_stream)
    d.tile_stats, d.in_tile_stats, d.in_stats_dil = _ptr(tile_stats), _ptr(in_tile_stats), int(in_stats_dil)
    d.in_gb, d.in_ld_gb = _ptr(in_gb), int(in_ld_gb)
    d.up_fir = _ptr(up_fir)
    d.pad_w_p1 = 0 if pad_w is None else int(pad_w) + 1
    d.rgb_only = int(rgb_only)
    d.in_absdiff = int(in_absdiff)
    if splitk_ws is not None:  # fp32 workspace tensor enabling split-K (see vt_conv2d_ws_bytes)
        d.splitk_ws, d.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel() * splitk_ws.element_size()
    return d


def conv2d(*, stream_of=None, **kw):
    d = make_conv_desc(**kw)
    t = stream_of
    if t is None:
        t = kw["out"] if isinstance(kw["out"], torch.Tensor) else kw["src0"]
    _lib.check(_lib.lib().vt_conv2d(C.byref(d), _stream(t)), "vt_conv2d")


def pack_conv_weight(w: torch.Tensor, cin_dst=None, chan_map=None, scale=1.0, src_transposed=False,
                     out_dtype=torch.float32):
    """w: (cout, cin, kh, kw) fp32 (or (cin, cout, kh, kw) when src_transposed)."""
    _dev_ok(w, chan_map)
    if src_transposed:
        cin_src, cout, kh, kw = w.shape
    else:
        cout, cin_src, kh, kw = w.shape
    cin_dst = cin_dst or (cin_src + 7) // 8 * 8
    out = torch.empty((cout, kh * kw, cin_dst), dtype=out_dtype, device=w.device)
    _lib.check(_lib.lib().vt_pack_conv_weight(_p(out), _p(w), cout, cin_src, kh, kw, cin_dst, _p(chan_map),
                                             float(scale), int(src_transposed), dt_code(out_dtype),
                                             _stream(w)), "vt_pack_conv_weight")
    return out


def conv_tile_stats_bytes(n, h, w, dil, c) -> int:
    return int(_lib.lib().vt_conv_tile_stats_bytes(n, h, w, dil, c))


def conv_weight_stream(packed: torch.Tensor):
    """packed [cout][taps][cin] (pack_conv_weight / modulate_weight layout) -> its fragment-stream image for
    the whole-K conv kernel (vt_conv_desc.weight_stream), or None when the shape has none."""
    _dev_ok(packed)
    cout, taps, cin = packed.shape
    dt = dt_code(packed.dtype)
    nbytes = int(_lib.lib().vt_conv_weight_stream_bytes(cout, taps, cin, dt))
    if nbytes <= 0:
        return None
    out = torch.empty((nbytes // packed.element_size(),), dtype=packed.dtype, device=packed.device)
    _lib.check(_lib.lib().vt_conv_weight_stream(_p(out), _p(packed), cout, taps, cin, dt, _stream(packed)),
               "vt_conv_weight_stream")
    return out


def modulate_weight(weight: torch.Tensor, s: torch.Tensor, scale: float, demodulate: bool, fir=None,
                    out=None, out_dtype=torch.float32):
    """weight (cout, cin, k, k) fp32, s (cin,) fp32 -> packed [phases*cout][k*k][cin]."""
    _dev_ok(weight, s, fir)
    cout, cin, k, _ = weight.shape
    phases = 4 if fir is not None else 1
    if out is None:
        out = torch.empty((phases * cout, (9 if fir is not None else k * k), cin), dtype=out_dtype,
                          device=weight.device)
    _lib.check(_lib.lib().vt_modulate_weight(_p(out), _p(weight), _p(s), cout, cin, k, float(scale),
                                            int(demodulate), _p(fir), dt_code(out.dtype), _stream(weight)),
               "vt_modulate_weight")
    return out


def linear(x, W, b=None, w_scale=1.0, b_scale=1.0, act=ACT_NONE, slope=0.2, gain=1.0, out=None,
           ld_x=None, ld_y=None, rows=None):
    """y = act(x @ W.T * w_scale + b * b_scale); x (rows, in_dim) fp32."""
    in_dim = W.shape[1]
    out_dim = W.shape[0]
    rows = rows if rows is not None else x.numel() // in_dim
    ld_x = ld_x or in_dim
    if out is None:
        out = torch.empty((rows, out_dim), dtype=torch.float32, device=W.device)
    ld_y = ld_y or out_dim
    _lib.check(_lib.lib().vt_linear(_p(out), ld_y, _p(x), ld_x, _p(W), _p(b), rows, in_dim, out_dim,
                                   float(w_scale), float(b_scale), act, float(slope), float(gain),
                                   _stream(W)), "vt_linear")
    return out


def pixel_norm(x):
    _dev_ok(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().vt_pixel_norm(_p(out), _p(x), x.shape[0], x.shape[1], _stream(x)), "vt_pixel_norm")
    return out


def instnorm_ws_bytes(n, hw, c_total):
    return int(_lib.lib().vt_instnorm_ws_bytes(n, hw, c_total))


def instnorm_stats(scale, shift, x, ld_x, n, hw, c, partials, dtype, other=None, ld_other=0,
                   style_gb=None, ld_gb=0, stream_of=None):
    def ptr(v):
        return C.c_void_p(0 if v is None else (v.data_ptr() if isinstance(v, torch.Tensor) else int(v)))
    t = stream_of if stream_of is not None else scale
    _lib.check(_lib.lib().vt_instnorm_stats(ptr(scale), ptr(shift), ptr(x), ld_x, ptr(other), ld_other, n, hw,
                                           c, ptr(style_gb), ld_gb, ptr(partials), dtype, _stream(t)),
               "vt_instnorm_stats")


def affine_apply(out, ld_out, x, ld_x, scale, shift, n, hw, c, dtype, other=None, ld_other=0,
                 stream_of=None):
    def ptr(v):
        return C.c_void_p(0 if v is None else (v.data_ptr() if isinstance(v, torch.Tensor) else int(v)))
    t = stream_of if stream_of is not None else scale
    _lib.check(_lib.lib().vt_affine_apply(ptr(out), ld_out, ptr(x), ld_x, ptr(other), ld_other, ptr(scale),
                                         ptr(shift), n, hw, c, dtype, _stream(t)), "vt_affine_apply")


def fusion_pack(out, ld_out, f_e, ld_e, mask, skip, n, hw, c, dtype, stream_of=None):
    def ptr(v):
        return C.c_void_p(0 if v is None else (v.data_ptr() if isinstance(v, torch.Tensor) else int(v)))
    t = stream_of if stream_of is not None else skip
    _lib.check(_lib.lib().vt_fusion_pack(ptr(out), ld_out, ptr(f_e), ld_e, ptr(mask), ptr(skip), n, hw, c,
                                        dtype, _stream(t)), "vt_fusion_pack")


def nchw_to_nhwc(x: torch.Tensor, out_dtype, ld_out=None, out=None):
    """(N,C,H,W) -> (N,H,W,ld_out) with channels >= C zero-filled up to a multiple of 8."""
    _dev_ok(x)
    n, c, h, w = x.shape
    cpad = (c + 7) // 8 * 8
    ld_out = ld_out or cpad
    if out is None:
        out = torch.empty((n, h, w, ld_out), dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().vt_nchw_to_nhwc(_p(out), ld_out, _p(x), n, c, h * w, dt_code(x.dtype),
                                         dt_code(out.dtype), _stream(x)), "vt_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x_ptr, ld_in, n, c, h, w, in_dtype, out_dtype, device, stream_of):
    out = torch.empty((n, c, h, w), dtype=out_dtype, device=device)
    ptr = x_ptr.data_ptr() if isinstance(x_ptr, torch.Tensor) else int(x_ptr)
    _lib.check(_lib.lib().vt_nhwc_to_nchw(_p(out), C.c_void_p(ptr), ld_in, n, c, h * w, dt_code(in_dtype),
                                         dt_code(out_dtype), _stream(stream_of)), "vt_nhwc_to_nchw")
    return out


def mfma_selftest(a: torch.Tensor, b: torch.Tensor):
    _dev_ok(a, b)
    c = torch.empty((16, 16), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().vt_mfma_selftest(_p(c), _p(a), _p(b), dt_code(a.dtype), _stream(a)),
               "vt_mfma_selftest")
    return c
