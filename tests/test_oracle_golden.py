"""Pins the CPU oracle (oracle/vtoonify_oracle.py) against tensors computed by the REAL
reference (tests/golden/make_golden.py).  CPU only.

Tolerances: upfirdn2d on integer-valued inputs with dyadic FIR taps is BIT-EXACT (every
partial sum is exactly representable, so summation order is irrelevant); everything
else is fp32 with a different summation order than torch's CPU kernels: 2e-5 relative
to the tensor's max-abs (measured ~1e-6).
"""
import numpy as np
import pytest

from conftest import load_golden, load_keys, rel_err
from oracle import vtoonify_oracle as O
from vtoonify_amd import synth

TOL = 2e-5


@pytest.fixture(params=["numpy", "torch"])
def backend(request):
    """Both evaluations of the dense contractions (oracle docstring) are pinned to the goldens."""
    old = O.set_backend(request.param)
    yield request.param
    O.set_backend(old)


def test_upfirdn2d_all_cases(backend):   # (both back-ends: the torch one is what cpu_baseline and every full-size comparison run)
    d, meta = load_golden("op_upfirdn2d.npz")
    assert len(meta) >= 12
    for m in meta:
        n = m["name"]
        up = tuple(m["up"]) if isinstance(m["up"], list) else m["up"]
        down = tuple(m["down"]) if isinstance(m["down"], list) else m["down"]
        y = O.upfirdn2d(d[n + "__x"], d[n + "__k"], up=up, down=down, pad=tuple(m["pad"]))
        ref = d[n + "__y"]
        assert y.shape == ref.shape, n
        if m["integer"]:
            assert np.array_equal(y, ref), f"{n}: not bit-exact"
        else:
            assert rel_err(y, ref) < 1e-6, n


def test_fused_leaky_relu_bit_exact(backend):
    d, meta = load_golden("op_fused_act.npz")
    for m in meta:
        n = m["name"]
        b = d.get(n + "__b")
        y = O.fused_leaky_relu(d[n + "__x"], b, m["slope"], m["scale"])
        assert np.array_equal(y, d[n + "__y"]), n


def _sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix) and "__" not in k}


def test_modules(backend):
    d, _ = load_golden("modules.npz")
    # StyledConv up / same
    for name, up in [("sc_up", True), ("sc_same", False)]:
        sd = _sub(d, name + ".")
        conv = O.modulated_conv2d(d[name + "__x"], d[name + "__s"], sd["conv.weight"],
                                  sd["conv.modulation.weight"], sd["conv.modulation.bias"],
                                  True, up, sd.get("conv.blur.kernel"))
        assert rel_err(conv, d[name + "__conv"]) < TOL
        y = O.styled_conv({name + "." + k: v for k, v in sd.items()}, name + ".", d[name + "__x"],
                          d[name + "__s"], up)
        assert rel_err(y, d[name + "__y"]) < TOL
    sd = {k: v for k, v in d.items() if k.startswith("rgb.")}
    y = O.to_rgb(sd, "rgb.", d["rgb__x"], d["rgb__s"], d["rgb__skip"])
    assert rel_err(y, d["rgb__y"]) < TOL
    assert rel_err(O.to_rgb(sd, "rgb.", d["rgb__x"], d["rgb__s"], None), d["rgb__y_noskip"]) < TOL
    # EqualLinear
    y = O.equal_linear(d["el_plain__x"], d["el_plain.weight"], d["el_plain.bias"], 1.0, False)
    assert rel_err(y, d["el_plain__y"]) < TOL
    y = O.equal_linear(d["el_act__x"], d["el_act.weight"], d["el_act.bias"], 0.01, True)
    assert rel_err(y, d["el_act__y"]) < TOL
    # AdaResBlock
    for dil in (1, 2, 4):
        n = f"ada_d{dil}"
        sd = {k: v for k, v in d.items() if k.startswith(n + ".")}
        y = O.ada_res_block(sd, n + ".", d[n + "__x"], d[n + "__s"], 0.7, dil)
        assert rel_err(y, d[n + "__y"]) < TOL, n
        assert np.array_equal(O.ada_res_block(sd, n + ".", d[n + "__x"], d[n + "__s"], 0, dil),
                              d[n + "__y0"])
    sd = {k: v for k, v in d.items() if k.startswith("vres.")}
    assert rel_err(O.vtoonify_res_block(sd, "vres.", d["vres__x"]), d["vres__y"]) < TOL
    sd = {k: v for k, v in d.items() if k.startswith("fus.")}
    f_out, m_e = O.fusion(sd, "fus.", d["fus__fg"], d["fus__fe"], 0.6)
    assert rel_err(f_out, d["fus__out"]) < TOL
    assert rel_err(m_e, d["fus__mask"]) < TOL
    k = synth.fir_kernel_2d().numpy()
    assert rel_err(O.upfirdn2d(d["mod_up__x"], k * 4, up=2, pad=tuple(d["mod_up__pad"])),
                   d["mod_up__y"]) < 1e-6
    assert rel_err(O.upfirdn2d(d["mod_up__x"], k, down=2, pad=tuple(d["mod_dn__pad"])),
                   d["mod_dn__y"]) < 1e-6


@pytest.mark.parametrize("tag,bb", [("D", "dualstylegan"), ("T", "toonify")])
def test_end_to_end(tag, bb, backend):
    d, _ = load_golden(f"e2e_{tag}.npz")
    shapes = load_keys(tag)
    assert len(shapes) == (399 if tag == "D" else 229)
    sd = synth.to_numpy_sd(synth.synth_state_dict(shapes, 0))
    x, s = d["x"], d["style"]
    for key in [k for k in d if k.startswith("y_ds")]:
        d_s = float(key[4:])
        y = O.vtoonify_forward(sd, x, s, d_s, bb)
        assert y.shape == (1, 3, 128, 128)
        assert rel_err(y, d[key]) < TOL, key
    feat, skip = O.vtoonify_forward(sd, x, s, 0.5, bb, return_feat=True)
    assert rel_err(feat, d["feat_ds0.5"]) < TOL and rel_err(skip, d["skip_ds0.5"]) < TOL
    if tag == "D":
        _, masks = O.vtoonify_forward(sd, x, s, 0.5, bb, return_mask=True)
        for i, m in enumerate(masks):
            assert rel_err(m, d[f"mask{i}_ds0.5"]) < TOL
    assert rel_err(O.vtoonify_forward(sd, x, s[:, 3], 0.5, bb), d["y_wspace"]) < TOL
    y2 = O.vtoonify_forward(sd, d["x2"], d["style2"], 0.75, bb)
    assert rel_err(y2, d["y2_ds0.75"]) < TOL
    assert rel_err(O.zplus2wplus(sd, d["zplus"], bb), d["wplus"]) < TOL


def test_end_to_end_mid_size_ragged():
    """The oracle against the REFERENCE at a mid-size, ragged geometry (tests/golden/e2e_D_mid.npz, make_golden.py::gen_e2e_mid):
    VToonify-D, batch 2, 72 x 104 (a 9 x 13-pixel trunk, dilated convs across image borders, levels that no tile divides),
    two styles.  The full-size GPU tests compare HIP with the oracle; this pins the oracle itself away from 32 x 32."""
    d, _ = load_golden("e2e_D_mid.npz")
    sd = synth.to_numpy_sd(synth.synth_state_dict(load_keys("D"), 0))
    sx, s0, s1 = (int(v) for v in d["seeds"])
    h, w = (int(v) for v in d["hw"])
    x = synth.synth_frames(2, h, w, seed=sx).numpy()
    s = np.concatenate([synth.synth_style(seed=s0).numpy(), synth.synth_style(seed=s1).numpy()], 0)
    old = O.set_backend("torch")
    try:
        y = O.vtoonify_forward(sd, x, s, 0.5, "dualstylegan")
    finally:
        O.set_backend(old)
    assert y.shape == d["y_ds05"].shape == (2, 3, 4 * h, 4 * w)
    assert rel_err(y, d["y_ds05"]) < TOL


def test_psp_encoder(backend):
    """oracle/psp_oracle.py vs the reference's GradualStyleEncoder (tests/golden/make_golden_psp.py)."""
    from oracle import psp_oracle as P
    d, _ = load_golden("psp.npz")
    shapes = load_keys("psp")
    assert len(shapes) == 621
    sd = synth.to_numpy_sd(synth.synth_state_dict(shapes, 0))
    for name in ("s32", "s64"):
        y, (c1, c2, c3, p2, p1) = P.gradual_style_encoder(sd, d[name + "__x"], return_taps=True)
        assert y.shape == d[name + "__y"].shape
        assert rel_err(c1, d[name + "__c1"]) < TOL and rel_err(c3, d[name + "__c3"]) < TOL, name
        assert rel_err(y, d[name + "__y"]) < TOL, name
