"""Backward passes of the drop-in operator surface against gradients of the REAL reference.

tests/golden/grads.npz (made by tests/golden/make_golden_grads.py from the reference's own StyledConv(upsample)
-> StyledConv -> ToRGB modules on its op_cpu operators) holds parameters, inputs, and three families of
gradients: first order, path-length-style second order (g_path_regularize, util.py:91-99) and R1-style second
order under no_weight_gradients() (d_r1_loss, util.py:75-82).  The graph is rebuilt HERE from the stored
tensors on vtoonify_amd.op (conv2d_gradfix.conv2d / conv_transpose2d with groups = batch, upfirdn2d,
fused_leaky_relu -- the calls ModulatedConv2d.forward makes, model/stylegan/model.py:259-306, restated), so
the check runs without the reference tree: on the host emulation AND on the MI355X (-m gpu), where the
double-backward kernels had no coverage in round 1."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from vtoonify_amd import op
from vtoonify_amd.op import conv2d_gradfix

TOL = 2e-4


def _modulated_conv(x, style, P, prefix, k, demodulate, upsample, blur_kernel):
    """ModulatedConv2d.forward, fused branch (model.py:259-306): modulation EqualLinear (scale 1/sqrt(style_dim),
    model.py:152-162), weight * style, demodulation, batch folded into groups."""
    W = P[prefix + "conv.weight"]                        # (1, cout, cin, k, k)
    mw, mb = P[prefix + "conv.modulation.weight"], P[prefix + "conv.modulation.bias"]
    B, cin, h, w = x.shape
    cout = W.shape[1]
    s = torch.nn.functional.linear(style, mw * (1.0 / math.sqrt(mw.shape[1])), mb).view(B, 1, cin, 1, 1)
    wt = (1.0 / math.sqrt(cin * k * k)) * W * s
    if demodulate:
        wt = wt * torch.rsqrt(wt.pow(2).sum([2, 3, 4]) + 1e-8).view(B, cout, 1, 1, 1)
    if upsample:
        xin = x.reshape(1, B * cin, h, w)
        wt = wt.transpose(1, 2).reshape(B * cin, cout, k, k)
        out = conv2d_gradfix.conv_transpose2d(xin, wt, padding=0, stride=2, groups=B)
        out = out.view(B, cout, out.shape[2], out.shape[3])
        return op.upfirdn2d(out, blur_kernel, pad=(1, 1))
    xin = x.reshape(1, B * cin, h, w)
    out = conv2d_gradfix.conv2d(xin, wt.view(B * cout, cin, k, k), padding=k // 2, groups=B)
    return out.view(B, cout, out.shape[2], out.shape[3])


def _styled_conv(x, style, noise, P, prefix, upsample, blur_kernel):
    out = _modulated_conv(x, style, P, prefix, 3, True, upsample, blur_kernel)
    out = out + P[prefix + "noise.weight"] * noise                       # NoiseInjection, model.py:315-320
    return op.fused_leaky_relu(out, P[prefix + "activate.bias"])         # FusedLeakyReLU, model.py:364-370


def test_reference_gradients_first_and_second_order(dev):
    d, _ = load_golden("grads.npz")
    T = lambda a: torch.from_numpy(a).to(dev)
    P = {k[3:]: T(v).requires_grad_(k != "p__up.conv.blur.kernel") for k, v in d.items() if k.startswith("p__")}
    x, s = T(d["x"]).requires_grad_(True), T(d["s"]).requires_grad_(True)
    proj, noise, kern = T(d["proj"]), T(d["noise"]), T(d["blur_kernel"])
    y = _styled_conv(x, s, noise, P, "up.", True, kern)
    y = _styled_conv(y, s, noise, P, "same.", False, None)
    img = _modulated_conv(y, s, P, "rgb.", 1, False, False, None) + P["rgb.bias"]       # ToRGB, model.py:383-392
    ref = d["img"]
    assert float(np.abs(img.detach().cpu().numpy() - ref).max() / np.abs(ref).max()) < 1e-5
    names = [k for k in P]
    params = [P[k] for k in names]
    g1 = torch.autograd.grad((img * proj).sum(), [x, s] + params, create_graph=True, allow_unused=True)
    g2 = torch.autograd.grad(g1[1].pow(2).sum(), [x] + params, retain_graph=True, allow_unused=True)
    with conv2d_gradfix.no_weight_gradients():
        gx, = torch.autograd.grad(img.sum(), [x], create_graph=True)
    g3 = torch.autograd.grad(gx.pow(2).sum(), params, allow_unused=True)
    checked = 0
    for tag, gs, nm in (("g1", g1, ["x", "s"] + names), ("g2", g2, ["x"] + names), ("g3", g3, names)):
        for k, g in zip(nm, gs):
            key = f"{tag}__{k}"
            if key not in d:
                continue
            assert g is not None, key
            want = d[key]
            err = float(np.abs(g.detach().cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-30))
            assert err < TOL, (key, err)
            checked += 1
    assert checked >= 40


def test_upfirdn2d_and_fused_leaky_relu_double_backward(dev):
    """Second derivatives of the two native ops (op/upfirdn2d.py:63-86, op/fused_act.py:40-84 in the reference)
    against torch autograd on CPU formulas: d/dx [ sum (d out / d x . v)^2 ]-style graphs."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 9, 7, generator=g)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kern = torch.outer(k1, k1) / 16.0
    v = torch.randn(2, 3, 18, 14, generator=g)

    def chain(xx, up_fn, act_fn, bias):
        y = act_fn(up_fn(xx.pow(2)), bias)        # x^2: the ops themselves are piecewise linear (zero Hessian)
        gx, = torch.autograd.grad((y * v.to(xx.device)).sum(), [xx], create_graph=True)
        return torch.autograd.grad(gx.pow(2).sum(), [xx])

    # CPU formulas: upfirdn2d(up=2, pad=(2,1)) = zero-insert + pad + conv with the flipped kernel
    def up_cpu(t):
        n, c, h, w = t.shape
        z = torch.zeros(n, c, 2 * h, 2 * w)
        z[:, :, ::2, ::2] = t
        z = torch.nn.functional.pad(z, (2, 1, 2, 1))
        wk = torch.flip(kern, [0, 1]).view(1, 1, 4, 4).repeat(c, 1, 1, 1)
        return torch.nn.functional.conv2d(z, wk, groups=c)

    def act_cpu(t, b):
        return torch.nn.functional.leaky_relu(t + b.view(1, -1, 1, 1), 0.2) * (2 ** 0.5)

    xc = x.clone().requires_grad_(True)
    bc = (0.1 * torch.randn(3, generator=g)).requires_grad_(True)
    want = chain(xc, up_cpu, act_cpu, bc)
    xd = x.to(dev).requires_grad_(True)
    bd = bc.detach().to(dev).requires_grad_(True)
    kd = kern.to(dev)
    got = chain(xd, lambda t: op.upfirdn2d(t, kd, up=2, pad=(2, 1)), lambda t, b: op.fused_leaky_relu(t, b), bd)
    for a, b in zip(got, want):
        err = float((a.cpu() - b).abs().max() / b.abs().max())
        assert err < 1e-4, err


@pytest.mark.parametrize("case", [
    # n, ci, co, h, w, k, stride, pad, dil, transposed, chunk bytes (None = one GEMM)
    (2, 8, 16, 9, 7, 3, 1, 1, 1, False, None),
    (2, 5, 6, 9, 7, 3, 1, 1, 1, False, None),            # channel counts that are no multiple of 8
    (3, 8, 8, 12, 10, 3, 2, 1, 1, False, None),           # stride 2 (encoder convs, vtoonify.py:167-176)
    (2, 8, 8, 12, 11, 3, 1, 2, 2, False, None),           # dilation 2 (AdaResBlock, dualstylegan.py:24-45)
    (2, 8, 8, 10, 10, 1, 1, 0, 1, False, None),           # 1x1 (ToRGB)
    (2, 8, 16, 6, 5, 3, 2, 0, 1, True, None),             # conv_transpose2d stride 2 (up-sampling StyledConv, model.py:273-286)
    (3, 8, 16, 9, 7, 3, 1, 1, 1, False, 20000),           # cut into groups of images
    (2, 8, 16, 9, 7, 3, 1, 1, 1, False, 3000),            # cut into groups of rows
])
def test_weight_gradient_contraction_vs_torch(dev, case, monkeypatch):
    """conv2d_gradfix's weight gradient (pixels as the contraction axis of one split-K GEMM, op/conv2d_gradfix.py:188-223 in
    the reference = cudnn_convolution_backward_weight) against torch autograd of F.conv2d / F.conv_transpose2d on the CPU,
    fp32 (bar 2e-5 relative to max|ref|) and bf16 tensors (2e-2, against the same gradient of the bf16-rounded operands)."""
    n, ci, co, h, w, k, s, p, d, transposed, chunk = case
    if chunk:
        monkeypatch.setattr(conv2d_gradfix, "_GW_CHUNK_BYTES", chunk)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g)
    F = torch.nn.functional
    if transposed:
        wt = torch.randn(ci, co, k, k, generator=g) * 0.1
        ref_fn = lambda a, b: F.conv_transpose2d(a, b, stride=s, padding=p, dilation=d)
        our_fn = lambda a, b: conv2d_gradfix.conv_transpose2d(a, b, stride=s, padding=p, dilation=d)
    else:
        wt = torch.randn(co, ci, k, k, generator=g) * 0.1
        ref_fn = lambda a, b: F.conv2d(a, b, stride=s, padding=p, dilation=d)
        our_fn = lambda a, b: conv2d_gradfix.conv2d(a, b, stride=s, padding=p, dilation=d)
    v = torch.randn(ref_fn(x, wt).shape, generator=g)
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        xr, wr = x.to(dtype).float().requires_grad_(True), wt.to(dtype).float().requires_grad_(True)
        vq = v.to(dtype).float()
        want, = torch.autograd.grad((ref_fn(xr, wr) * vq).sum(), [wr])
        xq, wq = x.to(dtype).to(dev).requires_grad_(True), wt.to(dtype).to(dev).requires_grad_(True)
        got, = torch.autograd.grad((our_fn(xq, wq).float() * vq.to(dev)).sum(), [wq])
        assert got.dtype == dtype and got.shape == wt.shape
        err = float((got.float().cpu() - want).abs().max() / want.abs().max())
        assert err < tol, (case, dtype, err)
