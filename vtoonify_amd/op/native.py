"""CPU-tensor branch of the operator surface: plain torch formulas, differentiable by autograd.

The reference's `model/stylegan/op` sends CPU tensors to a torch formula instead of its native kernels
(op/upfirdn2d.py:159-160 -> upfirdn2d_native, op/fused_act.py:105-116) and `conv2d_gradfix` falls through to
F.conv2d / F.conv_transpose2d whenever the input is not on a GPU (op/conv2d_gradfix.py:78-92); `style_transfer.py --cpu`
(:32,55) relies on it.  The formulas below are that contract written for this package:

    * a tensor on the GPU NEVER comes here -- it runs the gfx950 library or raises (kernels._dev_ok, _lib.lib());
    * nothing here imports `oracle/` (test infrastructure) or the reference.

upfirdn2d: zero-insertion by a strided write into a zero image, pads / crops by F.pad (negative values crop), the FIR as
one single-channel F.conv2d with the flipped taps over the planes, decimation by a strided view.
"""
import torch
import torch.nn.functional as F


def upfirdn2d(x, kernel, up, down, pad):
    """x (N, C, H, W); kernel (kh, kw); up / down (x, y); pad (x0, x1, y0, y1).  Output size as op/upfirdn2d.py:104-105."""
    n, c, h, w = x.shape
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    kh, kw = kernel.shape
    z = x.reshape(n * c, 1, h, w)
    if up_x > 1 or up_y > 1:
        u = z.new_zeros((n * c, 1, h * up_y, w * up_x))
        u[:, :, ::up_y, ::up_x] = z
        z = u
    if px0 or px1 or py0 or py1:
        z = F.pad(z, (px0, px1, py0, py1))
    if z.shape[2] < kh or z.shape[3] < kw:
        raise ValueError("upfirdn2d: empty output: pads crop away the whole image")
    taps = torch.flip(kernel, [0, 1]).to(dtype=z.dtype, device=z.device).reshape(1, 1, kh, kw)
    z = F.conv2d(z, taps)
    if down_x > 1 or down_y > 1:
        z = z[:, :, ::down_y, ::down_x]
    return z.reshape(n, c, z.shape[2], z.shape[3])


def fused_leaky_relu(x, bias, negative_slope, scale):
    """leaky_relu(x + bias[c], slope) * scale, bias on dim 1.  (The reference's CPU branch hard-codes the slope to 0.2,
    op/fused_act.py:110,116; its op_cpu twin and its GPU kernel honour the argument -- so does this; every caller passes 0.2.)"""
    if bias is not None:
        x = x + bias.reshape((1, -1) + (1,) * (x.ndim - 2)).to(x.dtype)
    return F.leaky_relu(x, negative_slope) * scale


def conv2d(x, weight, bias, stride, padding, dilation, groups):
    return F.conv2d(x, weight, bias, stride, padding, dilation, groups)


def conv_transpose2d(x, weight, bias, stride, padding, output_padding, groups, dilation):
    return F.conv_transpose2d(x, weight, bias, stride, padding, output_padding, groups, dilation)
