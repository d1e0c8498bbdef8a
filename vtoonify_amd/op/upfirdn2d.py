"""upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)) -- same contract as the reference
(model/stylegan/op/upfirdn2d.py:149-165): NCHW input, 2-D FIR `kernel`, `up`/`down` int or
(x, y), `pad` (p0, p1) or (x0, x1, y0, y1) with negative values cropping; returns a new
(N, C, out_h, out_w) tensor of the input's dtype.

CPU tensors take a plain torch formula (op/native.py), as in the reference (op/upfirdn2d.py:159-160);
GPU tensors run the gfx950 library or raise -- there is no fallback between the two.  fp64 inputs run
a double-arithmetic kernel (the reference dispatches double too, upfirdn2d_kernel.cu:311).
Differentiable: the gradient is
another upfirdn2d with the flipped kernel and swapped factors (op/upfirdn2d.py:20-61,
108-117), executed by the same HIP kernel.
"""
from collections import abc

import torch
from torch.autograd import Function

from .. import _lib
from .. import kernels as K
from . import native


def _planes(x, fir, up, down, pad):
    n, c, h, w = x.shape
    out = K.upfirdn2d_planes(x.reshape(n * c, h, w), fir, up[0], up[1], down[0], down[1], *pad)
    return out.view(n, c, out.shape[1], out.shape[2])


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, x, fir, up, down, pad):
        ctx.cfg = (up, down, pad, x.shape)
        ctx.save_for_backward(fir)
        return _planes(x.contiguous(), fir, up, down, pad)

    @staticmethod
    def backward(ctx, grad):
        (fir,) = ctx.saved_tensors
        up, down, pad, in_shape = ctx.cfg
        kh, kw = fir.shape
        in_h, in_w = in_shape[2], in_shape[3]
        out_h, out_w = grad.shape[2], grad.shape[3]
        # op/upfirdn2d.py:112-115
        gx0 = kw - pad[0] - 1
        gy0 = kh - pad[2] - 1
        gx1 = in_w * up[0] - out_w * down[0] + pad[0] - up[0] + 1
        gy1 = in_h * up[1] - out_h * down[1] + pad[2] - up[1] + 1
        g = _UpFirDn2d.apply(grad.contiguous(), torch.flip(fir, [0, 1]).contiguous(), down, up,
                             (gx0, gx1, gy0, gy1))
        return g, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if not isinstance(up, abc.Iterable):
        up = (up, up)
    if not isinstance(down, abc.Iterable):
        down = (down, down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    up, down, pad = tuple(int(v) for v in up), tuple(int(v) for v in down), tuple(int(v) for v in pad)
    if input.ndim != 4:
        raise ValueError("upfirdn2d expects an (N, C, H, W) tensor")
    if kernel.ndim != 2:
        raise ValueError("upfirdn2d expects a 2-D FIR kernel")
    if input.device.type == "cpu" and not _lib.emulation_injected():   # (the emulation: tests running the kernel sources on the host)
        return native.upfirdn2d(input, kernel.detach(), up, down, pad)
    fir = kernel.detach().to(device=input.device,
                             dtype=torch.float64 if input.dtype == torch.float64 else torch.float32).contiguous()
    return _UpFirDn2d.apply(input, fir, up, down, pad)
