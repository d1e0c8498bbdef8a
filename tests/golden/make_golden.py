#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the authoring container (needs /root/reference, read-only).  The
reference selects its CPU operator twin by editing import lines
(model/stylegan/op_cpu/readme.md:5-12); the runtime equivalent used here aliases
sys.modules['model.stylegan.op'] -> model.stylegan.op_cpu before importing the model
(SURVEY.md 8c).  Nothing is copied from the reference: only tensors it computes.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.json

Weights for the end-to-end fixtures are NOT stored: both sides regenerate them with
vtoonify_amd.synth.synth_tensor(key, shape, seed) (a pure function of the key name).
"""
import importlib
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")

sys.path.insert(0, REF)
import model.stylegan  # noqa: E402  (the reference's package)

_cpu = importlib.import_module("model.stylegan.op_cpu")
_gf = importlib.import_module("model.stylegan.op_cpu.conv2d_gradfix")
sys.modules["model.stylegan.op"] = _cpu
sys.modules["model.stylegan.op.conv2d_gradfix"] = _gf
_cpu.conv2d_gradfix = _gf

import numpy as np  # noqa: E402
import torch  # noqa: E402
from model.vtoonify import VToonify, VToonifyResBlock, Fusion  # noqa: E402
from model.stylegan.model import (ModulatedConv2d, StyledConv, ToRGB, EqualLinear,  # noqa: E402
                                  Blur, Upsample, Downsample)
from model.dualstylegan import AdaResBlock, AdaptiveInstanceNorm  # noqa: E402
from model.stylegan.op_cpu import upfirdn2d as ref_upfirdn2d, fused_leaky_relu as ref_flrelu  # noqa: E402

sys.path.append(REPO)
from vtoonify_amd import synth  # noqa: E402

torch.set_grad_enabled(False)


def npy(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# --------------------------------------------------------------------------------------
def gen_upfirdn2d():
    g = torch.Generator().manual_seed(11)
    k4 = synth.fir_kernel_2d()
    k1x12 = torch.tensor([[1., 3, 5, 7, 9, 11, 11, 9, 7, 5, 3, 1]]) / 64.0  # dyadic taps: sums exact in fp32
    k3x2 = torch.tensor([[1., 2], [3, 4], [5, 6]]) / 16.0  # asymmetric: catches a missing flip
    cases = [
        # name, shape, kernel, up, down, pad, integer-valued?
        ("blur_pad11", (2, 3, 17, 19), k4 * 4, 1, 1, (1, 1), True),
        ("blur_pad22", (1, 2, 9, 33), k4, 1, 1, (2, 2), True),
        ("up2_pad21", (1, 3, 16, 12), k4 * 4, 2, 1, (2, 1), True),
        ("down2_pad11", (2, 4, 18, 22), k4, 1, 2, (1, 1), True),
        ("down2_negpad", (1, 2, 20, 21), k4, 1, 2, (-1, 0, -2, 1), True),
        ("sep_1x12_up21", (1, 2, 10, 14), k1x12, (2, 1), 1, (6, 5, 0, 0), True),
        ("sep_12x1_down12", (1, 2, 30, 9), k1x12.t().contiguous(), 1, (1, 2), (0, 0, 5, 6), True),
        ("asym_3x2_up3_down2", (1, 2, 7, 6), k3x2, (3, 2), (2, 3), (2, 1, 1, 3), True),
        ("blur_float_65", (1, 8, 65, 65), k4 * 4, 1, 1, (1, 1), False),
        ("up2_float", (2, 3, 32, 40), k4 * 4, 2, 1, (2, 1), False),
        ("down2_float", (1, 5, 33, 47), k4, 1, 2, (1, 1), False),
        ("tiny_1x1", (1, 1, 1, 1), k4 * 4, 2, 1, (2, 1), True),
        ("single_row_out", (1, 1, 4, 9), k4, 1, 1, (0, 0, 0, 0), True),
    ]
    out = {}
    meta = []
    for name, shape, k, up, down, pad, integer in cases:
        if integer:
            x = torch.randint(-8, 9, shape, generator=g).float()
        else:
            x = torch.randn(shape, generator=g)
        y = ref_upfirdn2d(x, k, up=up, down=down, pad=pad)
        out[name + "__x"] = npy(x)
        out[name + "__k"] = npy(k)
        out[name + "__y"] = npy(y)
        meta.append(dict(name=name, up=up, down=down, pad=list(pad), integer=integer))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("op_upfirdn2d.npz", **out)


def gen_fused_act():
    g = torch.Generator().manual_seed(12)
    cases = [
        ("nchw", (2, 5, 7, 9), True, 0.2, 2 ** 0.5),
        ("nc", (18, 512), True, 0.2, 2 ** 0.5),
        ("nobias", (1, 3, 4, 4), False, 0.2, 2 ** 0.5),
        ("slope01_scale1", (2, 4, 3, 3), True, 0.1, 1.0),
        ("ncl", (2, 6, 10), True, 0.2, 2 ** 0.5),
        ("odd_tail", (1, 3, 1, 37), True, 0.2, 2 ** 0.5),
    ]
    out, meta = {}, []
    for name, shape, has_b, slope, scale in cases:
        x = torch.randn(shape, generator=g)
        b = torch.randn(shape[1], generator=g) if has_b else None
        y = ref_flrelu(x, b, slope, scale)
        out[name + "__x"] = npy(x)
        if has_b:
            out[name + "__b"] = npy(b)
        out[name + "__y"] = npy(y)
        meta.append(dict(name=name, has_bias=has_b, slope=slope, scale=scale))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("op_fused_act.npz", **out)


def _sd(mod, prefix):
    return {prefix + k: npy(v) for k, v in mod.state_dict().items()}


def gen_modules():
    """Small instances of every layer type on the hot path, weights stored."""
    out = {}
    torch.manual_seed(21)
    g = torch.Generator().manual_seed(22)

    # ModulatedConv2d / StyledConv / ToRGB  (per-sample styles, B=2)
    for name, cin, cout, up, hw in [("sc_up", 16, 8, True, (5, 7)), ("sc_same", 8, 16, False, (6, 5))]:
        m = StyledConv(cin, cout, 3, 32, upsample=up)
        m.conv.modulation.bias.data += torch.randn(cin) * 0.1
        m.activate.bias.data = torch.randn(cout) * 0.3
        x = torch.randn(2, cin, *hw, generator=g)
        s = torch.randn(2, 32, generator=g)
        noise = torch.zeros(2, 1, hw[0] * (2 if up else 1), hw[1] * (2 if up else 1))
        y = m(x, s, noise=noise)
        out.update(_sd(m, name + "."))
        out[name + "__x"], out[name + "__s"], out[name + "__y"] = npy(x), npy(s), npy(y)
        # raw modulated conv output too (before bias/act)
        out[name + "__conv"] = npy(m.conv(x, s))
    m = ToRGB(16, 32)
    m.bias.data = torch.randn(1, 3, 1, 1) * 0.3
    x = torch.randn(2, 16, 8, 6, generator=g)
    s = torch.randn(2, 32, generator=g)
    skip = torch.randn(2, 3, 4, 3, generator=g)
    out.update(_sd(m, "rgb."))
    out["rgb__x"], out["rgb__s"], out["rgb__skip"] = npy(x), npy(s), npy(skip)
    out["rgb__y"] = npy(m(x, s, skip))
    out["rgb__y_noskip"] = npy(m(x, s, None))

    # EqualLinear with and without activation
    for name, act, lr in [("el_plain", None, 1.0), ("el_act", "fused_lrelu", 0.01)]:
        m = EqualLinear(24, 40, lr_mul=lr, activation=act)
        m.bias.data = torch.randn(40) * (0.2 / lr)
        x = torch.randn(5, 24, generator=g)
        out.update(_sd(m, name + "."))
        out[name + "__x"], out[name + "__y"] = npy(x), npy(m(x))

    # AdaIN + AdaResBlock (dilation 1,2,4), style dim 32
    for d in (1, 2, 4):
        m = AdaResBlock(16, style_dim=32, dilation=d)
        m.conv[0].weight.data *= 100.0  # undo the near-zero init (dualstylegan.py:35-36)
        m.conv2[0].weight.data *= 100.0
        m.conv[1].bias.data = torch.randn(16) * 0.2
        m.conv2[1].bias.data = torch.randn(16) * 0.2
        x = torch.randn(2, 16, 11, 9, generator=g)
        s = torch.randn(2, 32, generator=g)
        name = f"ada_d{d}"
        out.update(_sd(m, name + "."))
        out[name + "__x"], out[name + "__s"] = npy(x), npy(s)
        out[name + "__y"] = npy(m(x, s, 0.7))
        out[name + "__y0"] = npy(m(x, s, 0))

    # VToonifyResBlock
    m = VToonifyResBlock(16)
    x = torch.randn(2, 16, 7, 8, generator=g)
    out.update(_sd(m, "vres."))
    out["vres__x"], out["vres__y"] = npy(x), npy(m(x))

    # Fusion
    m = Fusion(16, 16, 16)
    f_g = torch.randn(2, 16, 6, 9, generator=g)
    f_e = torch.randn(2, 16, 6, 9, generator=g)
    f_out, m_e = m(f_g, f_e, 0.6)
    out.update(_sd(m, "fus."))
    out["fus__fg"], out["fus__fe"] = npy(f_g), npy(f_e)
    out["fus__out"], out["fus__mask"] = npy(f_out), npy(m_e)

    # Blur / Upsample / Downsample modules (pads derived by the reference ctor)
    x = torch.randn(1, 4, 9, 9, generator=g)
    up = Upsample([1, 3, 3, 1])
    dn = Downsample([1, 3, 3, 1])
    out["mod_up__x"], out["mod_up__y"] = npy(x), npy(up(x))
    out["mod_dn__y"] = npy(dn(x))
    out["mod_up__pad"] = np.array(up.pad)
    out["mod_dn__pad"] = np.array(dn.pad)
    save("modules.npz", **out)


def gen_e2e():
    for bb, tag in [("dualstylegan", "D"), ("toonify", "T")]:
        torch.manual_seed(0)
        m = VToonify(backbone=bb).eval()
        shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
        with open(os.path.join(HERE, f"keys_{tag}.json"), "w") as f:
            json.dump(shapes, f, indent=0, sort_keys=True)
        m.load_state_dict(synth.synth_state_dict(shapes, seed=0))
        out = {}
        x = synth.synth_frames(1, 32, 32, seed=1234)
        s = synth.synth_style(seed=4321)
        out["x"], out["style"] = npy(x), npy(s)
        for d_s in ([0.0, 0.5, 1.0] if bb == "dualstylegan" else [0.5]):
            out[f"y_ds{d_s}"] = npy(m(x, s, d_s=d_s))
        feat, skip = m(x, s, d_s=0.5, return_feat=True)
        out["feat_ds0.5"], out["skip_ds0.5"] = npy(feat), npy(skip)
        if bb == "dualstylegan":
            _, masks = m(x, s, d_s=0.5, return_mask=True)
            for i, mk in enumerate(masks):
                out[f"mask{i}_ds0.5"] = npy(mk)
        # W-space style (B,512)
        out["y_wspace"] = npy(m(x, s[:, 3], d_s=0.5))
        # non-square, batch 2 with per-sample styles
        x2 = synth.synth_frames(2, 24, 40, seed=77)
        s2 = torch.cat([synth.synth_style(seed=5), synth.synth_style(seed=6)], 0)
        out["x2"], out["style2"] = npy(x2), npy(s2)
        out["y2_ds0.75"] = npy(m(x2, s2, d_s=0.75))
        # Z+ -> W+ mapping
        z = torch.randn(2, 18, 512, generator=torch.Generator().manual_seed(3))
        out["zplus"], out["wplus"] = npy(z), npy(m.zplus2wplus(z))
        save(f"e2e_{tag}.npz", **out)


def gen_e2e_mid():
    """Mid-size, ragged pin of the oracle: VToonify-D, batch 2, 72 x 104 (9 x 13-pixel trunk, levels whose
    sizes are not multiples of any tile), two different styles, d_s = 0.5 -- every full-size GPU comparison is HIP vs oracle,
    so the oracle itself is checked against the reference at a size where tile edges and dilation borders matter.  Only
    the inputs' seeds and the output are stored (x / style are regenerated from synth)."""
    torch.manual_seed(0)
    m = VToonify(backbone="dualstylegan").eval()
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth.synth_state_dict(shapes, seed=0))
    x = synth.synth_frames(2, 72, 104, seed=2024)
    s = torch.cat([synth.synth_style(seed=15), synth.synth_style(seed=16)], 0)
    y = m(x, s, d_s=0.5)
    save("e2e_D_mid.npz", y_ds05=npy(y).astype(np.float32), seeds=np.array([2024, 15, 16]), hw=np.array([72, 104]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["upfirdn2d", "fused_act", "modules", "e2e", "e2e_mid"]
    for w in which:
        globals()["gen_" + w]()
