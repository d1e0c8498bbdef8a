"""VToonify.forward as the reference's EAGER operator sequence over `vtoonify_amd.op`.

The frame executor (engine.py) fuses the path into ~90 launches; this module is the other end of the boundary: the
graph the reference itself builds -- one call of the operator surface per layer, torch for the glue -- written
functionally over a `g_ema` state_dict:

    ModulatedConv2d / StyledConv / ToRGB   model/stylegan/model.py:259-306, 364-370, 383-392
    Blur / Upsample                        model/stylegan/model.py:32-50, 74-90
    EqualLinear / EqualConv2d / PixelNorm  model/stylegan/model.py:17-18, 114-124, 152-162
    AdaptiveInstanceNorm / AdaResBlock     model/dualstylegan.py:6-21, 24-45
    VToonifyResBlock / Fusion / forward    model/vtoonify.py:92-128, 210-277

Two users:
  * `VToonify` whose parameters are on the CPU (`style_transfer.py --cpu`, :32,55): the operators take their CPU-tensor
    branch (op/native.py) and this module IS the forward pass -- the reference's contract "CPU tensors -> native path";
  * the `-m gpu` drop-in test (tests/test_eager_graph.py): the same graph with GPU tensors runs every dense contraction,
    FIR and bias-activation through the gfx950 library, i.e. what the reference's model code does once its
    `model.stylegan.op` is this package.
It never reaches for the executor's fused kernels and imports nothing from `oracle/`.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .op import conv2d_gradfix, fused_leaky_relu, upfirdn2d

N_LATENT = 18
_DILATION = {1: 4, 2: 4, 3: 2, 4: 2, 5: 1, 6: 1}    # VToonify.res[i], model/vtoonify.py:201-207


def _equal_linear(x, w, b, lr_mul=1.0, activate=False):
    scale = (1.0 / math.sqrt(w.shape[1])) * lr_mul
    if activate:
        return fused_leaky_relu(F.linear(x, w * scale), b * lr_mul)
    return F.linear(x, w * scale, b * lr_mul)


def _pixel_norm(x):
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


class EagerVToonify:
    """sd: the reference's state_dict (any float dtype the operators take; every tensor on one device)."""

    def __init__(self, sd: Dict[str, torch.Tensor], backbone: str = "dualstylegan", in_size: int = 256):
        assert backbone in ("dualstylegan", "toonify")
        self.sd, self.backbone, self.in_size = sd, backbone, in_size
        self.dual = backbone == "dualstylegan"
        self.g = "generator.generator." if self.dual else "generator."
        self.n_down = 0
        while f"encoder.{self.n_down}.0.weight" in sd:
            self.n_down += 1

    # ------------------------------------------------------------------ style path
    def _mapping(self, prefix, z):
        """PixelNorm + EqualLinear(lr_mul 0.01, fused_lrelu) x n (model.py:411-420, dualstylegan.py:51-55)."""
        x = _pixel_norm(z)
        i = 1
        while f"{prefix}{i}.weight" in self.sd:
            x = _equal_linear(x, self.sd[f"{prefix}{i}.weight"], self.sd[f"{prefix}{i}.bias"], 0.01, True)
            i += 1
        return x

    def zplus2wplus(self, zplus):
        n, l, d = zplus.shape
        return self._mapping(self.g + "style.", zplus.reshape(n * l, d)).reshape(n, l, d)

    # ------------------------------------------------------------------ generator blocks
    def _modulated(self, x, style, name, k, demodulate, upsample):
        sd = self.sd
        wt = sd[name + "weight"]                                 # (1, cout, cin, k, k)
        b, cin, h, w = x.shape
        cout = wt.shape[1]
        s = _equal_linear(style, sd[name + "modulation.weight"], sd[name + "modulation.bias"]).view(b, 1, cin, 1, 1)
        wt = (1.0 / math.sqrt(cin * k * k)) * wt * s
        if demodulate:
            wt = wt * torch.rsqrt(wt.pow(2).sum([2, 3, 4]) + 1e-8).view(b, cout, 1, 1, 1)
        xin = x.reshape(1, b * cin, h, w)
        if upsample:
            wt = wt.transpose(1, 2).reshape(b * cin, cout, k, k)
            y = conv2d_gradfix.conv_transpose2d(xin, wt, padding=0, stride=2, groups=b)
            y = y.view(b, cout, y.shape[2], y.shape[3])
            return upfirdn2d(y, sd[name + "blur.kernel"], pad=(1, 1))           # Blur, model.py:74-90 (pad from :192-198)
        y = conv2d_gradfix.conv2d(xin, wt.reshape(b * cout, cin, k, k), padding=k // 2, groups=b)
        return y.view(b, cout, y.shape[2], y.shape[3])

    def _styled(self, x, style, name, upsample):
        y = self._modulated(x, style, name + "conv.", 3, True, upsample)
        # NoiseInjection with the zero noise of vtoonify.py:267 adds exactly 0; kept as the reference's operation order
        y = y + self.sd[name + "noise.weight"] * y.new_zeros(y.shape[0], 1, y.shape[2], y.shape[3])
        return fused_leaky_relu(y, self.sd[name + "activate.bias"])

    def _to_rgb(self, x, style, name, skip):
        y = self._modulated(x, style, name + "conv.", 1, False, False) + self.sd[name + "bias"]
        if skip is not None:
            y = y + upfirdn2d(skip, self.sd[name + "upsample.kernel"], up=2, down=1, pad=(2, 1))   # Upsample, model.py:32-50
        return y

    def _adain(self, x, style, name):
        gb = F.linear(style, self.sd[name + "style.weight"], self.sd[name + "style.bias"])[:, :, None, None]
        gamma, beta = gb.chunk(2, 1)
        return gamma * F.instance_norm(x, eps=1e-5) + beta

    def _conv_layer(self, x, name, dil):
        w = self.sd[name + "0.weight"]
        y = conv2d_gradfix.conv2d(x, w * (1.0 / math.sqrt(w.shape[1] * 9)), padding=dil, dilation=dil)
        return fused_leaky_relu(y, self.sd[name + "1.bias"])

    def _ada_res(self, x, s, d_s, i):
        if d_s == 0:
            return x
        name = f"res.{i}."
        y = self._conv_layer(self._adain(x, s, name + "norm."), name + "conv.", _DILATION[i])
        y = self._conv_layer(self._adain(y, s, name + "norm2."), name + "conv2.", _DILATION[i])
        return y * d_s + x

    def _conv(self, x, name, stride=1, pad=1):
        return conv2d_gradfix.conv2d(x, self.sd[name + "weight"], self.sd[name + "bias"], stride=stride, padding=pad)

    # ------------------------------------------------------------------ forward
    def forward(self, x, style, d_s=None, return_mask=False, return_feat=False):
        sd, g = self.sd, self.g
        if self.dual and d_s is None:
            raise TypeError("VToonify-D needs a style degree d_s (model/vtoonify.py:124)")
        b = x.shape[0]
        if style.ndim < 3:
            ada = style.unsqueeze(1).repeat(1, N_LATENT, 1)
            res = self._mapping("generator.style.", style).unsqueeze(1).repeat(1, N_LATENT, 1) if self.dual else None
        else:
            n, l, d = style.shape
            ada = style
            res = self._mapping("generator.style.", style.reshape(n * l, d)).reshape(n, l, d) if self.dual else None
        if ada.shape[0] == 1 and b > 1:
            ada = ada.expand(b, -1, -1)
            res = res.expand(b, -1, -1) if res is not None else None
        if self.dual:
            ada = ada.clone()
            for i in range(7, N_LATENT):
                ada[:, i] = _equal_linear(ada[:, i], sd[f"generator.res.{i}.weight"], sd[f"generator.res.{i}.bias"])
        # content encoder
        feat, feats = x, []
        for bi in range(self.n_down):
            st = 1 if bi == 0 else 2
            feat = F.leaky_relu(self._conv(feat, f"encoder.{bi}.0.", st), 0.2)
            feat = F.leaky_relu(self._conv(feat, f"encoder.{bi}.2."), 0.2)
            feats.append(feat)
        feats = feats[::-1]
        rk = f"encoder.{self.n_down}."
        for ii in range(6):
            y = F.leaky_relu(self._conv(feat, f"{rk}{ii}.conv."), 0.2)
            y = F.leaky_relu(self._conv(y, f"{rk}{ii}.conv2."), 0.2)
            feat = (y + feat) / math.sqrt(2)
            if self.dual:
                feat = self._ada_res(feat, res[:, ii + 1], d_s, ii + 1)
        out = feat
        skip = self._conv(feat, f"encoder.{self.n_down + 1}.", pad=0)
        if return_feat:
            return out, skip
        masks = []
        for lvl in range(5):
            if 2 ** (5 + lvl) <= self.in_size:
                f_e = feats[lvl]
                p = f"fusion_out.{lvl}."
                if self.dual:
                    lab = out.new_zeros(b, 1) + d_s
                    lab = F.leaky_relu(F.linear(lab, sd[p + "linear.0.weight"], sd[p + "linear.0.bias"]), 0.2)
                    lab = F.leaky_relu(F.linear(lab, sd[p + "linear.2.weight"], sd[p + "linear.2.bias"]), 0.2)
                    m = self._adain(torch.cat([out, (out - f_e).abs()], 1), lab, p + "norm.")
                    m = torch.tanh(F.relu(self._conv(m, p + "conv2.")))
                    out = self._conv(torch.cat([out, f_e * m], 1), p + "conv.")
                    skip = self._conv(torch.cat([skip, f_e * m], 1), f"fusion_skip.{lvl}.")
                    masks.append(m)
                else:
                    out_in = torch.cat([out, f_e], 1)
                    skip = self._conv(torch.cat([skip, f_e], 1), f"fusion_skip.{lvl}.")
                    out = self._conv(out_in, p)
            out = self._styled(out, ada[:, 2 * lvl + 7], f"{g}convs.{6 + 2 * lvl}.", True)
            out = self._styled(out, ada[:, 2 * lvl + 8], f"{g}convs.{7 + 2 * lvl}.", False)
            skip = self._to_rgb(out, ada[:, 2 * lvl + 9], f"{g}to_rgbs.{3 + lvl}.", skip)
        if return_mask and self.dual:
            return skip, masks
        return skip
