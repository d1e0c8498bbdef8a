"""conv2d_gradfix with the reference's surface (model/stylegan/op/conv2d_gradfix.py:8-75):
module globals `enabled` / `weight_gradients_disabled`, `no_weight_gradients()`,
`conv2d(...)`, `conv_transpose2d(...)`.

On the reference every call ends in cuDNN via F.conv2d / F.conv_transpose2d; here GPU
tensors run the MFMA implicit-GEMM kernel of libvtoonify_amd.so (fp32 inputs use the exact
fp32 MFMA, bf16 inputs the bf16 MFMA); CPU tensors end in F.conv2d / F.conv_transpose2d like the
reference's (its could_use_op() is False off the GPU, op/conv2d_gradfix.py:78-92).  `groups` (the per-sample trick of ModulatedConv2d,
model.py:273-304) is a loop of launches.

Autograd follows the reference's structure (op/conv2d_gradfix.py:134-223), every contraction on the
same HIP kernel family:
  grad_input   = the opposite convolution of grad_output with the weight (conv <-> conv_transpose,
                 output_padding from the shapes, :122-132), itself differentiable;
  grad_weight  = Conv2dGradWeight (honours `weight_gradients_disabled`): the pixels are the contraction
                 axis -- dW = A B^T with A = the kh*kw shifted views of the input, B = grad_output, K = N*Ho*Wo,
                 as one split-K 1x1 contraction (_grad_weight_kernel) -- where the reference calls
                 cudnn_convolution_backward_weight; its backward gives the second-order terms
                 (R1 / path-length regularisers, util.py:75-82);
  grad_bias    = grad_output.sum((0, 2, 3)).
"""
import contextlib
import ctypes as C

import torch

from .. import _lib
from .. import kernels as K
from . import native

enabled = True
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    yield
    weight_gradients_disabled = old


def _one(v, what):
    if isinstance(v, (tuple, list)):
        if len(set(int(i) for i in v)) != 1:
            raise NotImplementedError(f"{what} must be the same on both axes")
        return int(v[0])
    return int(v)


def _pair(v, what):
    if isinstance(v, (tuple, list)):
        if len(v) != 2:
            raise ValueError(f"{what} must be an int or a pair")
        return int(v[0]), int(v[1])
    return int(v), int(v)


def _check(input, weight):
    if input.ndim != 4 or weight.ndim != 4:
        raise ValueError("expected 4-D input and weight")
    if input.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError("conv2d_gradfix supports fp32 and bf16 inputs")


def _launch(input, weight, bias, stride, padding, dilation, groups, transposed, output_padding):
    """One forward contraction (no autograd): F.conv2d / F.conv_transpose2d semantics."""
    _check(input, weight)
    stride, padding, dilation = _one(stride, "stride"), _one(padding, "padding"), _one(dilation, "dilation")
    oph, opw = _pair(output_padding, "output_padding")
    dtype = input.dtype
    n, cin, h, w = input.shape
    if transposed:
        cin_w, cout_g, kh, kw = weight.shape
        if cin_w != cin:
            raise ValueError("conv_transpose2d: weight.shape[0] must equal input channels")
        cout = cout_g * groups
        out_h = (h - 1) * stride - 2 * padding + dilation * (kh - 1) + oph + 1
        out_w = (w - 1) * stride - 2 * padding + dilation * (kw - 1) + opw + 1
    else:
        cout, cin_g, kh, kw = weight.shape
        if cin_g * groups != cin:
            raise ValueError("conv2d: weight.shape[1] * groups must equal input channels")
        cout_g = cout // groups
        out_h = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
        out_w = (w + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    if out_h <= 0 or out_w <= 0:
        raise ValueError("convolution output would be empty")
    cin_g = cin // groups
    cpad = (cin_g + 7) // 8 * 8
    w32 = weight.detach().to(torch.float32).contiguous()
    b32 = bias.detach().to(torch.float32).contiguous() if bias is not None else None
    x = input.detach().contiguous()
    # conv_transpose2d with stride 1 IS a convolution with the kernel turned by 180 degrees and the channel axes swapped
    # (padding d*(k-1) - p): it takes the patch kernels of the forward pass instead of the gather form -- this is the
    # grad_input of every stride-1 convolution (op/conv2d_gradfix.py:147-160 in the reference).  Data movement on the weight only.
    as_conv = transposed and stride == 1 and oph == 0 and opw == 0 and dilation * (kh - 1) - padding >= 0 and kh == kw
    conv_pad = dilation * (kh - 1) - padding if as_conv else padding
    # wide outputs: NHWC out of the fast epilogues + one tiled layout change (vt_nhwc_to_nchw); narrow ones (ToRGB, masks)
    # keep the planar output of the thin kernels
    via_nhwc = cout_g >= 32 and cout_g % 8 == 0
    # (the NHWC path of an ungrouped conv returns its own tensor: nothing is allocated here for it -- ADVICE r5)
    out = None if (via_nhwc and groups == 1) else \
        torch.empty((n, cout, out_h, out_w), dtype=(dtype if via_nhwc else torch.float32), device=input.device)
    for g in range(groups):
        xg = x[:, g * cin_g:(g + 1) * cin_g].contiguous() if groups > 1 else x
        x_nhwc = K.nchw_to_nhwc(xg, dtype, ld_out=cpad)
        if transposed:
            wg = w32[g * cin_g:(g + 1) * cin_g].contiguous() if groups > 1 else w32
            if as_conv:
                wp = K.pack_conv_weight(wg.flip(2, 3).transpose(0, 1).contiguous(), cin_dst=cpad, out_dtype=dtype)
            else:
                wp = K.pack_conv_weight(wg, cin_dst=cpad, src_transposed=True, out_dtype=dtype)
        else:
            wg = w32[g * cout_g:(g + 1) * cout_g].contiguous() if groups > 1 else w32
            wp = K.pack_conv_weight(wg, cin_dst=cpad, out_dtype=dtype)
        bg = b32[g * cout_g:(g + 1) * cout_g].contiguous() if b32 is not None else None
        common = dict(src0=x_nhwc, c0=cpad, ld0=cpad, n=n, h=h, w=w, out_h=out_h, out_w=out_w, weight=wp, cout=cout_g, kh=kh,
                      kw=kw, stride=stride, pad=conv_pad, dil=dilation, transposed=int(transposed and not as_conv), bias=bg,
                      dtype=K.dt_code(dtype))
        if via_nhwc:
            o_nhwc = torch.empty((n, out_h, out_w, cout_g), dtype=dtype, device=input.device)
            K.conv2d(out=o_nhwc, ld_out=cout_g, **common)
            og = K.nhwc_to_nchw(o_nhwc, cout_g, n, cout_g, out_h, out_w, dtype, dtype, input.device, o_nhwc)
        else:
            og = out if groups == 1 else torch.empty((n, cout_g, out_h, out_w), dtype=torch.float32, device=input.device)
            K.conv2d(out=og, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32, **common)
        if groups > 1:
            out[:, g * cout_g:(g + 1) * cout_g] = og
        elif via_nhwc:
            out = og
    return out if out.dtype == dtype else out.to(dtype)


def _output_padding(cfg, input_shape, output_shape, weight_shape):
    """output_padding of the opposite convolution that maps grad_output back to the input's size
    (op/conv2d_gradfix.py:122-132)."""
    transposed, stride, padding, _, dilation = cfg
    if transposed:
        return (0, 0)
    return tuple(input_shape[i + 2] - (output_shape[i + 2] - 1) * stride - (1 - 2 * padding)
                 - dilation * (weight_shape[i + 2] - 1) for i in range(2))


_GW_CHUNK_BYTES = 1 << 28   # one im2col operand per GEMM launch stays below this (transient memory of a backward conv: the A
#                             operand, the fp32 copy of grad_output the weight packer reads and its packed form -- ADVICE r5: 1 GiB
#                             chunks put 2-4 GB of transients on a 128-channel 256^2 layer; the 32-bit buffer ranges of the kernels
#                             allow far more)


def _grad_weight_kernel(inp, grad, kh, kw, stride, padding, dilation):
    """dW[c_grad, c_inp, ky, kx] = sum_{n,oy,ox} grad[n,c_grad,oy,ox] * inp[n,c_inp,oy*s+ky*d-p,ox*s+kx*d-p]
    (what the reference gets from cudnn_convolution_backward_weight, op/conv2d_gradfix.py:188-223).

    The PIXELS are the contraction axis: with A[(tap, c_inp)][(n,oy,ox)] = the tap's shifted, strided view of the
    zero-padded input (kh*kw strided device copies -- data movement only) and B[c_grad][(n,oy,ox)] = grad, dW = A B^T is a
    GEMM with K = N*Ho*Wo.  It runs as a 1x1 convolution of the MFMA kernels over an "image" of kh*kw*C_inp pixels with
    N*Ho*Wo channels, split along K over the whole GPU (fp32 slices summed in slice order: deterministic), fp32 out.
    (Round 1-4 form: the batch as the channel axis of a conv with an Ho x Wo "filter" -- 4 of 32 K lanes of every MFMA
    used and one K step per filter tap: 11.3 ms for a 128->128 3x3 layer at 4 x 256^2 in bf16, profiles/r05_grad_bench.json.)
    Batches too large for one operand are cut into groups of images / rows; their fp32 partial results are added."""
    n, ca, h, w = inp.shape
    n2, cb, ho, wo = grad.shape
    s, p, d = stride, padding, dilation
    if n2 != n or (ho - 1) * s + (kh - 1) * d + 1 > h + 2 * p or (wo - 1) * s + (kw - 1) * d + 1 > w + 2 * p:
        raise ValueError("conv2d_gradfix: inconsistent shapes in the weight gradient")
    dtype, dev = inp.dtype, inp.device
    esz = 4 if dtype == torch.float32 else 2
    taps = kh * kw
    m = taps * ca
    xp = torch.nn.functional.pad(inp.detach(), (p, p, p, p)).transpose(0, 1)     # (Ca, N, Hp, Wp) view
    gt = grad.detach().transpose(0, 1)                                            # (Cb, N, Ho, Wo) view
    # groups of (images, output rows) whose operands fit the chunk size
    per_row = max(m * esz, cb * (4 + esz)) * wo          # A rows, or grad_output's fp32 copy + its packed form
    if per_row * ho <= _GW_CHUNK_BYTES:
        step = max(1, _GW_CHUNK_BYTES // (per_row * ho))
        groups = [(i, min(n, i + step), 0, ho) for i in range(0, n, step)]
    else:
        rstep = max(1, _GW_CHUNK_BYTES // per_row)
        groups = [(i, i + 1, y, min(ho, y + rstep)) for i in range(n) for y in range(0, ho, rstep)]
    lib = _lib.lib()
    acc = None
    for n0, n1, y0, y1 in groups:
        ni, rows = n1 - n0, y1 - y0
        k = ni * rows * wo
        kp = (k + 63) // 64 * 64                                                  # whole K steps of either dtype
        a = torch.empty((taps, ca, kp), dtype=dtype, device=dev)
        if kp != k:
            a[:, :, k:].zero_()
        for ky in range(kh):
            for kx in range(kw):
                r0, c0 = ky * d + y0 * s, kx * d
                a[ky * kw + kx, :, :k].unflatten(1, (ni, rows, wo)).copy_(
                    xp[:, n0:n1, r0:r0 + s * (rows - 1) + 1:s, c0:c0 + s * (wo - 1) + 1:s])
        b = gt[:, n0:n1, y0:y1].reshape(cb, k, 1, 1).to(torch.float32).contiguous()
        wp = K.pack_conv_weight(b, cin_dst=kp, out_dtype=dtype)
        out = torch.empty((1, cb, 1, m), dtype=torch.float32, device=dev)
        kwargs = dict(src0=a, c0=kp, ld0=kp, n=1, h=1, w=m, out_h=1, out_w=m, weight=wp, cout=cb, kh=1, kw=1, out=out,
                      ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32, dtype=K.dt_code(dtype))
        need = int(lib.vt_conv2d_ws_bytes(C.byref(K.make_conv_desc(**kwargs))))
        if need < 0:
            raise _lib.VtError(f"vt_conv2d descriptor rejected: {lib.vt_last_error().decode()}")
        ws = torch.zeros((need,), dtype=torch.uint8, device=dev) if need else None
        K.conv2d(splitk_ws=ws, **kwargs)
        acc = out if acc is None else acc + out
    dw = acc.view(cb, kh, kw, ca).permute(0, 3, 1, 2).contiguous()
    return dw if dtype == torch.float32 else dw.to(dtype)


class _Conv(torch.autograd.Function):
    """groups == 1.  cfg = (transposed, stride, padding, output_padding, dilation)."""

    @staticmethod
    def forward(ctx, input, weight, bias, cfg):
        transposed, stride, padding, output_padding, dilation = cfg
        ctx.cfg = cfg
        ctx.save_for_backward(input, weight)
        return _launch(input, weight, bias, stride, padding, dilation, 1, transposed, output_padding)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        transposed, stride, padding, _, dilation = ctx.cfg
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            p = _output_padding(ctx.cfg, input.shape, grad_output.shape, weight.shape)
            grad_input = _Conv.apply(grad_output, weight, None, (not transposed, stride, padding, p, dilation))
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            grad_weight = _ConvGradWeight.apply(grad_output, input, ctx.cfg, tuple(weight.shape))
        if ctx.needs_input_grad[2]:
            grad_bias = grad_output.sum((0, 2, 3))
        return grad_input, grad_weight, grad_bias, None


class _ConvGradWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, cfg, weight_shape):
        transposed, stride, padding, _, dilation = cfg
        ctx.cfg, ctx.weight_shape = cfg, weight_shape
        ctx.save_for_backward(grad_output, input)
        kh, kw = weight_shape[2], weight_shape[3]
        g, x = grad_output.detach(), input.detach()
        if transposed:   # weight is (Cin, Cout, kh, kw): the roles of input and grad_output swap
            return _grad_weight_kernel(g, x, kh, kw, stride, padding, dilation)
        return _grad_weight_kernel(x, g, kh, kw, stride, padding, dilation)

    @staticmethod
    def backward(ctx, grad_grad_weight):
        grad_output, input = ctx.saved_tensors
        transposed, stride, padding, _, dilation = ctx.cfg
        gg_output = gg_input = None
        if ctx.needs_input_grad[0]:
            gg_output = _Conv.apply(input, grad_grad_weight, None, ctx.cfg)
        if ctx.needs_input_grad[1]:
            p = _output_padding(ctx.cfg, input.shape, grad_output.shape, ctx.weight_shape)
            gg_input = _Conv.apply(grad_output, grad_grad_weight, None, (not transposed, stride, padding, p, dilation))
        return gg_output, gg_input, None, None


def _run(input, weight, bias, stride, padding, dilation, groups, transposed, output_padding):
    if input.device.type == "cpu" and not _lib.emulation_injected():
        # the reference's could_use_op() is False off the GPU: plain F.conv2d / F.conv_transpose2d (op/conv2d_gradfix.py:78-92)
        if transposed:
            return native.conv_transpose2d(input, weight, bias, stride, padding, output_padding, groups, dilation)
        return native.conv2d(input, weight, bias, stride, padding, dilation, groups)
    _check(input, weight)
    needs_grad = torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad or
                                              (bias is not None and bias.requires_grad))
    if not needs_grad:
        return _launch(input, weight, bias, stride, padding, dilation, groups, transposed, output_padding)
    stride, padding, dilation = _one(stride, "stride"), _one(padding, "padding"), _one(dilation, "dilation")
    cfg = (bool(transposed), stride, padding, _pair(output_padding, "output_padding"), dilation)
    if groups == 1:
        return _Conv.apply(input, weight, bias, cfg)
    # grouped convolution (ModulatedConv2d folds the batch into groups, model.py:273-304): one
    # differentiable launch per group; the slicing / concatenation is data movement only
    xs = input.chunk(groups, dim=1)
    ws = weight.chunk(groups, dim=0)
    bs = bias.chunk(groups, dim=0) if bias is not None else [None] * groups
    return torch.cat([_Conv.apply(x.contiguous(), w.contiguous(), b, cfg) for x, w, b in zip(xs, ws, bs)], dim=1)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return _run(input, weight, bias, stride, padding, dilation, groups, False, 0)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return _run(input, weight, bias, stride, padding, dilation, groups, True, output_padding)
