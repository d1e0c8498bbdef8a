"""CPU oracle for the VToonify per-frame hot path.  TEST INFRASTRUCTURE ONLY.

A plain-numpy (fp32) restatement of the reference algorithm, written from the
reference's own CPU path (model/stylegan/op_cpu) and model code; every function cites
the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; nothing under vtoonify_amd/ does.

Parity pinning: the reference ships no tests / golden vectors, so this
oracle is pinned against outputs of the reference ITSELF, produced in the authoring
container by tests/golden/make_golden.py (which imports /root/reference with
model.stylegan.op aliased to model.stylegan.op_cpu) and committed under
tests/golden/*.npz; tests/test_oracle_golden.py checks every fixture.

Third-party arithmetic: the reference's dense contractions bottom out in
torch.nn.functional.{conv2d,conv_transpose2d,linear,instance_norm} (PyTorch, pinned
1.7.1 in environment/vtoonify_env.yaml:22; 2.10.0 in this image).  They are restated
here from their published definitions (cross-correlation, zero padding; biased
variance, eps inside the sqrt).

All tensors are numpy fp32, NCHW, exactly the reference's layout.

Backends for the three dense contractions (conv2d / conv_transpose2d / linear) and, since round 4, for upfirdn2d,
fused_leaky_relu, leaky_relu and instance_norm (the reference's op_cpu path runs all of them through torch on CPU threads:
with only the contractions on torch the oracle needed 3.3x the reference's time per frame, profiles/r04_cpu_port_vs_reference.txt):
  "numpy"  (default) the plain restatement below -- one BLAS matmul per filter tap;
  "torch"  the very functions the reference calls, torch.nn.functional.conv2d /
           conv_transpose2d / linear on CPU tensors (op/conv2d_gradfix.py:34-42,66-75,
           model/stylegan/model.py:154-160).  ~100x faster at the 22x256x256 benchmark size,
           which the numpy form needs minutes for; tests/test_oracle_golden.py pins both
           backends against the same golden vectors and against each other.  Everything else
           (upfirdn2d, fused_leaky_relu, modulation, AdaIN, the graph) is numpy either way.
Select with set_backend(); full-size GPU parity tests and bench.py's cpu_baseline use "torch".
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
SQRT2 = F32(2.0 ** 0.5)
BACKEND = "numpy"


def set_backend(name: str) -> str:
    """Choose how conv2d / conv_transpose2d / linear are evaluated; returns the old setting."""
    global BACKEND
    if name not in ("numpy", "torch"):
        raise ValueError(name)
    old, BACKEND = BACKEND, name
    return old


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=F32))


# --------------------------------------------------------------------------------------
# operator surface (model/stylegan/op_cpu)
# --------------------------------------------------------------------------------------
def _pair(v):
    if isinstance(v, (tuple, list)):
        return int(v[0]), int(v[1])
    return int(v), int(v)


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Upsample (zero insertion) -> pad/crop -> FIR -> decimate.

    Follows op_cpu/upfirdn2d.py:7-60 (`upfirdn2d` + `upfirdn2d_native`):
    up/down are ints or (x, y) tuples; pad is (p0, p1) for both axes or
    (x0, x1, y0, y1); negative pads crop; the FIR is applied as a correlation with the
    FLIPPED kernel (true convolution); out = (in*up + pad0 + pad1 - k + down) // down.
    """
    x = np.asarray(x, dtype=F32)
    kernel = np.asarray(kernel, dtype=F32)
    up_x, up_y = _pair(up)
    down_x, down_y = _pair(down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    px0, px1, py0, py1 = (int(p) for p in pad)
    n, c, in_h, in_w = x.shape
    kh, kw = kernel.shape
    if BACKEND == "torch":
        # the same steps through the torch functions op_cpu/upfirdn2d.py:27-58 itself calls (F.pad for the zero insertion
        # and the positive pads, slicing for the negative ones, F.conv2d with the flipped kernel, strided slicing): what the
        # reference's CPU path executes, multi-threaded -- bench.py's cpu_baseline times this form
        import torch
        import torch.nn.functional as TF
        t = _t(x).reshape(n * c, in_h, 1, in_w, 1)
        t = TF.pad(t, [0, up_x - 1, 0, 0, 0, up_y - 1]).reshape(n * c, 1, in_h * up_y, in_w * up_x)
        t = TF.pad(t, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
        t = t[:, :, max(-py0, 0): t.shape[2] - max(-py1, 0), max(-px0, 0): t.shape[3] - max(-px1, 0)]
        if t.shape[2] < kh or t.shape[3] < kw:
            oh = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
            ow = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
            return np.zeros((n, c, max(oh, 0), max(ow, 0)), dtype=F32)
        t = TF.conv2d(t, torch.flip(_t(kernel), [0, 1])[None, None])
        t = t[:, :, ::down_y, ::down_x]
        return np.ascontiguousarray(t.reshape(n, c, t.shape[2], t.shape[3]).numpy())

    # zero insertion (op_cpu/upfirdn2d.py:30-32)
    z = np.zeros((n, c, in_h * up_y, in_w * up_x), dtype=F32)
    z[:, :, ::up_y, ::up_x] = x
    # positive pads, then negative pads as crops (op_cpu/upfirdn2d.py:34-43)
    z = np.pad(z, ((0, 0), (0, 0), (max(py0, 0), max(py1, 0)), (max(px0, 0), max(px1, 0))))
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]

    full_h = in_h * up_y + py0 + py1 - kh + 1
    full_w = in_w * up_x + px0 + px1 - kw + 1
    out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
    out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
    if full_h <= 0 or full_w <= 0:
        return np.zeros((n, c, max(out_h, 0), max(out_w, 0)), dtype=F32)
    kf = kernel[::-1, ::-1]  # torch.flip(kernel,[0,1]) op_cpu/upfirdn2d.py:49
    acc = np.zeros((n, c, full_h, full_w), dtype=F32)
    for ky in range(kh):
        for kx in range(kw):
            acc += z[:, :, ky:ky + full_h, kx:kx + full_w] * kf[ky, kx]
    out = acc[:, :, ::down_y, ::down_x]  # op_cpu/upfirdn2d.py:57
    assert out.shape[2] == out_h and out.shape[3] == out_w, (out.shape, out_h, out_w)
    return np.ascontiguousarray(out)


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    """leaky_relu(x + b[c]) * scale, bias broadcast on dim 1 (op_cpu/fused_act.py:23-34)."""
    x = np.asarray(x, dtype=F32)
    if BACKEND == "torch":   # F.leaky_relu(input + bias) * scale, op_cpu/fused_act.py:27-34
        import torch.nn.functional as TF
        t = _t(x)
        if bias is not None:
            t = t + _t(bias).reshape((1, -1) + (1,) * (x.ndim - 2))
        return (TF.leaky_relu(t, negative_slope) * F32(scale)).numpy()
    if bias is not None:
        b = np.asarray(bias, dtype=F32).reshape((1, -1) + (1,) * (x.ndim - 2))
        x = x + b
    y = np.where(x >= 0, x, x * F32(negative_slope)).astype(F32)
    return (y * F32(scale)).astype(F32)


# --------------------------------------------------------------------------------------
# third-party contractions restated (torch.nn.functional)
# --------------------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, padding=0, dilation=1):
    """Cross-correlation, zero padding, groups=1 (F.conv2d as reached from
    op/conv2d_gradfix.py:34-42 and the nn.Conv2d modules of model/vtoonify.py)."""
    x = np.asarray(x, dtype=F32)
    w = np.asarray(w, dtype=F32)
    if BACKEND == "torch":
        import torch.nn.functional as TF
        return TF.conv2d(_t(x), _t(w), None if b is None else _t(b), stride=stride, padding=padding,
                         dilation=dilation).numpy()
    n, cin, h, wd = x.shape
    cout, cin2, kh, kw = w.shape
    assert cin == cin2
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding)))
    ho = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    wo = (wd + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    out = np.zeros((n, cout, ho * wo), dtype=F32)
    for ky in range(kh):
        for kx in range(kw):
            y0, x0 = ky * dilation, kx * dilation
            xs = xp[:, :, y0:y0 + (ho - 1) * stride + 1:stride, x0:x0 + (wo - 1) * stride + 1:stride]
            xs = np.ascontiguousarray(xs).reshape(n, cin, ho * wo)
            out += np.matmul(w[:, :, ky, kx][None], xs)
    out = out.reshape(n, cout, ho, wo)
    if b is not None:
        out = out + np.asarray(b, dtype=F32).reshape(1, -1, 1, 1)
    return out.astype(F32)


def conv_transpose2d(x, w, stride=2):
    """F.conv_transpose2d, padding 0, weight (Cin, Cout, kh, kw): every input pixel
    scatters its kh x kw patch (model/stylegan/model.py:281-283)."""
    x = np.asarray(x, dtype=F32)
    w = np.asarray(w, dtype=F32)
    if BACKEND == "torch":
        import torch.nn.functional as TF
        return TF.conv_transpose2d(_t(x), _t(w), stride=stride).numpy()
    n, cin, h, wd = x.shape
    cin2, cout, kh, kw = w.shape
    assert cin == cin2
    ho = (h - 1) * stride + kh
    wo = (wd - 1) * stride + kw
    out = np.zeros((n, cout, ho, wo), dtype=F32)
    xf = x.reshape(n, cin, h * wd)
    for ky in range(kh):
        for kx in range(kw):
            contrib = np.matmul(w[:, :, ky, kx].T[None], xf).reshape(n, cout, h, wd)
            out[:, :, ky:ky + (h - 1) * stride + 1:stride, kx:kx + (wd - 1) * stride + 1:stride] += contrib
    return out


def linear(x, w, b=None):
    if BACKEND == "torch":
        import torch.nn.functional as TF
        return TF.linear(_t(x), _t(w), None if b is None else _t(b)).numpy()
    y = np.asarray(x, dtype=F32) @ np.asarray(w, dtype=F32).T
    if b is not None:
        y = y + np.asarray(b, dtype=F32)
    return y.astype(F32)


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d(affine=False): per (n,c) biased variance (dualstylegan.py:10)."""
    x = np.asarray(x, dtype=F32)
    if BACKEND == "torch":
        import torch.nn.functional as TF
        return TF.instance_norm(_t(x), eps=eps).numpy()
    mean = x.mean(axis=(2, 3), keepdims=True, dtype=np.float64)
    var = ((x - mean) ** 2).mean(axis=(2, 3), keepdims=True, dtype=np.float64)
    return ((x - mean) / np.sqrt(var + eps)).astype(F32)


def leaky_relu(x, slope=0.2):
    if BACKEND == "torch":
        import torch.nn.functional as TF
        return TF.leaky_relu(_t(x), slope).numpy()
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


# --------------------------------------------------------------------------------------
# StyleGAN2 layers (model/stylegan/model.py)
# --------------------------------------------------------------------------------------
def pixel_norm(x):
    """model/stylegan/model.py:17-18."""
    x = np.asarray(x, dtype=F32)
    return (x / np.sqrt(np.mean(x * x, axis=1, keepdims=True) + F32(1e-8))).astype(F32)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """EqualLinear.forward (model/stylegan/model.py:152-162)."""
    scale = F32((1 / math.sqrt(weight.shape[1])) * lr_mul)
    w = np.asarray(weight, dtype=F32) * scale
    if activation:
        out = linear(x, w)
        return fused_leaky_relu(out, np.asarray(bias, dtype=F32) * F32(lr_mul))
    return linear(x, w, np.asarray(bias, dtype=F32) * F32(lr_mul))


def modulated_weight(weight, style_row, mod_w, mod_b, demodulate=True):
    """Per-sample modulated (and demodulated) weight (model/stylegan/model.py:259-267).

    weight (1,Cout,Cin,k,k); style_row (512,) -> (Cout,Cin,k,k)."""
    _, cout, cin, k, _ = weight.shape
    s = equal_linear(style_row[None], mod_w, mod_b)[0]  # (Cin,), bias_init=1 lives in mod_b
    scale = F32(1 / math.sqrt(cin * k * k))
    w = scale * np.asarray(weight[0], dtype=F32) * s.reshape(1, cin, 1, 1)
    if demodulate:
        d = 1.0 / np.sqrt((w.astype(F32) ** 2).sum(axis=(1, 2, 3)) + F32(1e-8))
        w = w * d.reshape(cout, 1, 1, 1).astype(F32)
    return w.astype(F32)


def modulated_conv2d(x, style, weight, mod_w, mod_b, demodulate=True, upsample=False,
                     blur_kernel=None):
    """ModulatedConv2d.forward, fused branch (model/stylegan/model.py:259-306).
    The reference folds batch into groups; that is a per-sample loop."""
    x = np.asarray(x, dtype=F32)
    k = weight.shape[-1]
    outs = []
    for b in range(x.shape[0]):
        w = modulated_weight(weight, style[b], mod_w, mod_b, demodulate)
        if upsample:
            # conv_transpose2d stride 2 pad 0 with weight transposed to (Cin,Cout,k,k)
            # (model/stylegan/model.py:273-285), then Blur pad (1,1) for k=3 (:192-198)
            o = conv_transpose2d(x[b:b + 1], w.transpose(1, 0, 2, 3), stride=2)
            factor = 2
            p = (blur_kernel.shape[0] - factor) - (k - 1)
            pad0 = (p + 1) // 2 + factor - 1
            pad1 = p // 2 + 1
            o = upfirdn2d(o, blur_kernel, pad=(pad0, pad1))
        else:
            o = conv2d(x[b:b + 1], w, padding=k // 2)
        outs.append(o)
    return np.concatenate(outs, 0)


def styled_conv(sd, prefix, x, style, upsample):
    """StyledConv.forward (model/stylegan/model.py:364-370) with the zero inference
    noise of model/vtoonify.py:267 (NoiseInjection adds weight*0)."""
    out = modulated_conv2d(
        x, style, sd[prefix + "conv.weight"], sd[prefix + "conv.modulation.weight"],
        sd[prefix + "conv.modulation.bias"], demodulate=True, upsample=upsample,
        blur_kernel=sd.get(prefix + "conv.blur.kernel"))
    return fused_leaky_relu(out, sd[prefix + "activate.bias"])


def to_rgb(sd, prefix, x, style, skip):
    """ToRGB.forward (model/stylegan/model.py:383-392); Upsample pad=(2,1), kernel*4
    (model/stylegan/model.py:32-50)."""
    out = modulated_conv2d(x, style, sd[prefix + "conv.weight"],
                           sd[prefix + "conv.modulation.weight"],
                           sd[prefix + "conv.modulation.bias"], demodulate=False)
    out = out + sd[prefix + "bias"]
    if skip is not None:
        out = out + upfirdn2d(skip, sd[prefix + "upsample.kernel"], up=2, down=1, pad=(2, 1))
    return out.astype(F32)


# --------------------------------------------------------------------------------------
# DualStyleGAN pieces (model/dualstylegan.py)
# --------------------------------------------------------------------------------------
def adain(sd, prefix, x, style):
    """AdaptiveInstanceNorm.forward (model/dualstylegan.py:16-21)."""
    st = linear(style, sd[prefix + "style.weight"], sd[prefix + "style.bias"])
    c = x.shape[1]
    gamma = st[:, :c].reshape(-1, c, 1, 1)
    beta = st[:, c:].reshape(-1, c, 1, 1)
    return (gamma * instance_norm(x) + beta).astype(F32)


def conv_layer(sd, prefix, x, dilation):
    """ConvLayer = EqualConv2d(no bias, padding=dilation) + FusedLeakyReLU
    (model/stylegan/model.py:593-637, 114-124)."""
    w = sd[prefix + "0.weight"]
    scale = F32(1 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3]))
    out = conv2d(x, w * scale, None, stride=1, padding=dilation, dilation=dilation)
    return fused_leaky_relu(out, sd[prefix + "1.bias"])


def ada_res_block(sd, prefix, x, s, w, dilation):
    """AdaResBlock.forward (model/dualstylegan.py:38-45)."""
    if w == 0:
        return x
    out = conv_layer(sd, prefix + "conv.", adain(sd, prefix + "norm.", x, s), dilation)
    out = conv_layer(sd, prefix + "conv2.", adain(sd, prefix + "norm2.", out, s), dilation)
    return (out * F32(w) + x).astype(F32)


# --------------------------------------------------------------------------------------
# VToonify (model/vtoonify.py)
# --------------------------------------------------------------------------------------
def vtoonify_res_block(sd, prefix, x):
    """VToonifyResBlock.forward (model/vtoonify.py:100-104)."""
    out = leaky_relu(conv2d(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], padding=1))
    out = leaky_relu(conv2d(out, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], padding=1))
    return ((out + x) / F32(math.sqrt(2))).astype(F32)


def fusion(sd, prefix, f_g, f_e, d_s):
    """Fusion.forward (model/vtoonify.py:122-128)."""
    n = f_g.shape[0]
    lab = np.zeros((n, 1), dtype=F32) + F32(d_s)
    lab = leaky_relu(linear(lab, sd[prefix + "linear.0.weight"], sd[prefix + "linear.0.bias"]))
    lab = leaky_relu(linear(lab, sd[prefix + "linear.2.weight"], sd[prefix + "linear.2.bias"]))
    out = np.concatenate([f_g, np.abs(f_g - f_e)], 1)
    m = conv2d(adain(sd, prefix + "norm.", out, lab), sd[prefix + "conv2.weight"],
               sd[prefix + "conv2.bias"], padding=1)
    m_e = np.tanh(np.maximum(m, 0)).astype(F32)
    f_out = conv2d(np.concatenate([f_g, f_e * m_e], 1), sd[prefix + "conv.weight"],
                   sd[prefix + "conv.bias"], padding=1)
    return f_out, m_e


def mapping_network(sd, prefix, z, n_layers):
    """Sequential(PixelNorm, n x EqualLinear(lr_mul=.01, fused_lrelu))
    (model/stylegan/model.py:411-420; model/dualstylegan.py:51-55)."""
    out = pixel_norm(z)
    for i in range(1, n_layers + 1):
        out = equal_linear(out, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"],
                           lr_mul=0.01, activation=True)
    return out


def zplus2wplus(sd, zplus, backbone="dualstylegan"):
    """VToonify.zplus2wplus (model/vtoonify.py:285-286)."""
    g = "generator.generator." if backbone == "dualstylegan" else "generator."
    z = np.asarray(zplus, dtype=F32)
    return mapping_network(sd, g + "style.", z.reshape(-1, z.shape[-1]), 8).reshape(z.shape)


_DILATIONS = {1: 4, 2: 4, 3: 2, 4: 2, 5: 1, 6: 1}  # model/vtoonify.py:201-207


def vtoonify_forward(sd, x, style, d_s=None, backbone="dualstylegan", in_size=256,
                     return_mask=False, return_feat=False):
    """VToonify.forward (model/vtoonify.py:210-277).  sd: {reference key: numpy array}."""
    x = np.asarray(x, dtype=F32)
    style = np.asarray(style, dtype=F32)
    dual = backbone == "dualstylegan"
    g = "generator.generator." if dual else "generator."
    n_latent = 18

    # style mapping (vtoonify.py:212-224)
    if style.ndim < 3:
        if dual:
            resstyles = np.repeat(mapping_network(sd, "generator.style.", style, 2)[:, None], n_latent, 1)
        adastyles = np.repeat(style[:, None], n_latent, 1)
    else:
        nb, nl, nd = style.shape
        if dual:
            resstyles = mapping_network(sd, "generator.style.", style.reshape(nb * nl, nd), 2).reshape(nb, nl, nd)
        adastyles = style
    if dual:
        adastyles = adastyles.copy()
        for i in range(7, n_latent):
            adastyles[:, i] = equal_linear(adastyles[:, i], sd[f"generator.res.{i}.weight"],
                                           sd[f"generator.res.{i}.bias"])

    # content encoder (vtoonify.py:226-242; layers built at :160-183)
    feat = x
    feats = []
    n_down = int(math.log2(in_size)) - 4  # encoder_res = [256,128,64,32] -> 4 blocks incl. stem
    for bi in range(n_down):
        stride = 1 if bi == 0 else 2
        feat = leaky_relu(conv2d(feat, sd[f"encoder.{bi}.0.weight"], sd[f"encoder.{bi}.0.bias"],
                                 stride=stride, padding=1))
        feat = leaky_relu(conv2d(feat, sd[f"encoder.{bi}.2.weight"], sd[f"encoder.{bi}.2.bias"],
                                 padding=1))
        feats.append(feat)
    feats = feats[::-1]
    for ii in range(6):
        feat = vtoonify_res_block(sd, f"encoder.{n_down}.{ii}.", feat)
        if dual:
            feat = ada_res_block(sd, f"res.{ii + 1}.", feat, resstyles[:, ii + 1], d_s,
                                 _DILATIONS[ii + 1])
    out = feat
    skip = conv2d(feat, sd[f"encoder.{n_down + 1}.weight"], sd[f"encoder.{n_down + 1}.bias"])
    if return_feat:
        return out, skip

    # generator levels (vtoonify.py:245-272)
    m_es = []
    _index = 1
    for lvl in range(5):
        ci = 6 + 2 * lvl
        if 2 ** (5 + ((_index - 1) // 2)) <= in_size:
            fi = (_index - 1) // 2
            f_e = feats[fi]
            if dual:
                out, m_e = fusion(sd, f"fusion_out.{fi}.", out, f_e, d_s)
                skip = conv2d(np.concatenate([skip, f_e * m_e], 1), sd[f"fusion_skip.{fi}.weight"],
                              sd[f"fusion_skip.{fi}.bias"], padding=1)
                m_es.append(m_e)
            else:
                out = conv2d(np.concatenate([out, f_e], 1), sd[f"fusion_out.{fi}.weight"],
                             sd[f"fusion_out.{fi}.bias"], padding=1)
                skip = conv2d(np.concatenate([skip, f_e], 1), sd[f"fusion_skip.{fi}.weight"],
                              sd[f"fusion_skip.{fi}.bias"], padding=1)
        out = styled_conv(sd, f"{g}convs.{ci}.", out, adastyles[:, _index + 6], upsample=True)
        out = styled_conv(sd, f"{g}convs.{ci + 1}.", out, adastyles[:, _index + 7], upsample=False)
        skip = to_rgb(sd, f"{g}to_rgbs.{3 + lvl}.", out, adastyles[:, _index + 8], skip)
        _index += 2
    if return_mask and dual:
        return skip, m_es
    return skip
