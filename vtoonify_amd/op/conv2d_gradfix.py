"""conv2d_gradfix with the reference's surface (model/stylegan/op/conv2d_gradfix.py:8-75):
module globals `enabled` / `weight_gradients_disabled`, `no_weight_gradients()`,
`conv2d(...)`, `conv_transpose2d(...)`.

On the reference every call ends in cuDNN via F.conv2d / F.conv_transpose2d; here GPU
tensors run the MFMA implicit-GEMM kernel of libvtoonify_amd.so (fp32 inputs use the exact
fp32 MFMA, bf16 inputs the bf16 MFMA).  Forward only: tensors that require grad are
rejected (training is out of scope for this path).  `groups` (the
per-sample trick of ModulatedConv2d, model.py:273-304) is a loop of launches.
"""
import contextlib

import torch

from .. import kernels as K

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


def _check(input, weight):
    if torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad):
        raise NotImplementedError("vtoonify_amd.op.conv2d_gradfix is inference-only (no autograd)")
    if input.ndim != 4 or weight.ndim != 4:
        raise ValueError("expected 4-D input and weight")
    if input.dtype not in (torch.float32, torch.bfloat16):
        raise NotImplementedError("conv2d_gradfix supports fp32 and bf16 inputs")


def _run(input, weight, bias, stride, padding, dilation, groups, transposed, output_padding):
    _check(input, weight)
    stride, padding, dilation = _one(stride, "stride"), _one(padding, "padding"), _one(dilation, "dilation")
    output_padding = _one(output_padding, "output_padding")
    dtype = input.dtype
    n, cin, h, w = input.shape
    if transposed:
        cin_w, cout_g, kh, kw = weight.shape
        if cin_w != cin:
            raise ValueError("conv_transpose2d: weight.shape[0] must equal input channels")
        cout = cout_g * groups
        out_h = (h - 1) * stride - 2 * padding + dilation * (kh - 1) + output_padding + 1
        out_w = (w - 1) * stride - 2 * padding + dilation * (kw - 1) + output_padding + 1
        # gather form: iy = (oy + p_eff - ky*dil) / stride with p_eff = padding
        pad_eff = padding
    else:
        cout, cin_g, kh, kw = weight.shape
        if cin_g * groups != cin:
            raise ValueError("conv2d: weight.shape[1] * groups must equal input channels")
        cout_g = cout // groups
        out_h = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
        out_w = (w + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
        pad_eff = padding
    cin_g = cin // groups
    cpad = (cin_g + 7) // 8 * 8
    out = torch.empty((n, cout, out_h, out_w), dtype=torch.float32, device=input.device)
    w32 = weight.detach().to(torch.float32).contiguous()
    b32 = bias.detach().to(torch.float32).contiguous() if bias is not None else None
    x = input.detach().contiguous()
    for g in range(groups):
        xg = x[:, g * cin_g:(g + 1) * cin_g].contiguous() if groups > 1 else x
        x_nhwc = K.nchw_to_nhwc(xg, dtype, ld_out=cpad)
        if transposed:
            wg = w32[g * cin_g:(g + 1) * cin_g].contiguous() if groups > 1 else w32
            wp = K.pack_conv_weight(wg, cin_dst=cpad, src_transposed=True, out_dtype=dtype)
        else:
            wg = w32[g * cout_g:(g + 1) * cout_g].contiguous() if groups > 1 else w32
            wp = K.pack_conv_weight(wg, cin_dst=cpad, out_dtype=dtype)
        og = out if groups == 1 else torch.empty((n, cout_g, out_h, out_w), dtype=torch.float32,
                                                 device=input.device)
        K.conv2d(src0=x_nhwc, c0=cpad, ld0=cpad, n=n, h=h, w=w, out_h=out_h, out_w=out_w, weight=wp,
                 cout=cout_g, kh=kh, kw=kw, stride=stride, pad=pad_eff, dil=dilation,
                 transposed=int(transposed), bias=(b32[g * cout_g:(g + 1) * cout_g].contiguous()
                                                   if b32 is not None else None),
                 out=og, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32, dtype=K.dt_code(dtype))
        if groups > 1:
            out[:, g * cout_g:(g + 1) * cout_g] = og
    return out if dtype == torch.float32 else out.to(dtype)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return _run(input, weight, bias, stride, padding, dilation, groups, False, 0)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return _run(input, weight, bias, stride, padding, dilation, groups, True, output_padding)
