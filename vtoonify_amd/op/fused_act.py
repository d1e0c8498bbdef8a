"""FusedLeakyReLU / fused_leaky_relu with the reference's surface
(model/stylegan/op/fused_act.py:87-119): leaky_relu(x + bias[c], slope) * scale, bias
broadcast on dim 1, any rank >= 2, new tensor returned.  CPU tensors take the torch formula
of op/native.py like the reference's own CPU branch (op/fused_act.py:105-116); GPU tensors run
the gfx950 library or raise.  fp64 runs a double-arithmetic kernel (fused_bias_act_kernel.cu:96
dispatches double too).  Backward uses the same kernel in its grad mode
(fused_bias_act_kernel.cu:55-57 semantics), like op/fused_act.py:20-71.
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib
from .. import kernels as K
from . import native


class _FusedLeakyReLUBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, has_bias, slope, scale):
        ctx.save_for_backward(out)
        ctx.cfg = (slope, scale)
        grad_input = K.fused_bias_act(grad_output.contiguous(), None, out, 3, 1, slope, scale)
        grad_bias = None
        if has_bias:
            dims = [0] + list(range(2, grad_input.ndim))
            grad_bias = grad_input.sum(dims).detach()
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gg_input, gg_bias):
        (out,) = ctx.saved_tensors
        slope, scale = ctx.cfg
        x = gg_input.contiguous()
        b = gg_bias.to(x.dtype).contiguous() if gg_bias is not None else None
        return K.fused_bias_act(x, b, out, 3, 1, slope, scale), None, None, None, None


class _FusedLeakyReLU(Function):
    @staticmethod
    def forward(ctx, x, bias, slope, scale):
        out = K.fused_bias_act(x, bias, None, 3, 0, slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (bias is not None, slope, scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        has_bias, slope, scale = ctx.cfg
        gi, gb = _FusedLeakyReLUBackward.apply(grad_output, out, has_bias, slope, scale)
        return gi, (gb if has_bias else None), None, None


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if input.device.type == "cpu" and not _lib.emulation_injected():
        if bias is not None and (input.ndim < 2 or bias.numel() != input.shape[1]):
            raise ValueError("bias must have input.shape[1] elements")
        return native.fused_leaky_relu(input, bias, float(negative_slope), float(scale))
    x = input.contiguous()  # op/fused_act.py:119
    b = None
    if bias is not None:
        b = bias.to(device=x.device, dtype=x.dtype).contiguous()
        if x.ndim < 2 or b.numel() != x.shape[1]:
            raise ValueError("bias must have input.shape[1] elements")
    return _FusedLeakyReLU.apply(x, b, float(negative_slope), float(scale))


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
