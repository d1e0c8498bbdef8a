"""The other side of the drop-in boundary: the reference's EAGER graph over `vtoonify_amd.op` (vtoonify_amd/eager.py).

Three things are pinned here.

1. CPU tensors take the native torch branch of the operator surface (op/native.py) like the reference's own
   `model/stylegan/op` (op/upfirdn2d.py:159-165, op/fused_act.py:105-116, op/conv2d_gradfix.py:78-92): the three
   operators against the goldens the REAL reference produced (tests/golden/op_*.npz), `VToonify` constructed on the CPU
   (`style_transfer.py --cpu`, :32,55) against the reference-made 32x32 frames of e2e_*.npz.  No library is involved:
   the tests release the host emulation first.
2. `-m gpu`: the same graph with GPU tensors -- every contraction, FIR and bias-activation of ModulatedConv2d /
   StyledConv / ToRGB / AdaResBlock / Fusion (model/stylegan/model.py:259-306,364-392, model/dualstylegan.py:24-45,
   model/vtoonify.py:106-128) runs through the gfx950 library, one operator call per layer as the reference's model code
   issues them -- against the reference-made goldens at 32x32 and against the oracle at 256x256.
3. fp64 tensors through `upfirdn2d` / `fused_leaky_relu` (the reference's native ops dispatch double,
   upfirdn2d_kernel.cu:311, fused_bias_act_kernel.cu:96): bit-exact on the integer cases, double arithmetic elsewhere.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, load_keys, rel_err
from vtoonify_amd import _lib, op, synth
from vtoonify_amd.eager import EagerVToonify
from vtoonify_amd.op import conv2d_gradfix
from vtoonify_amd.vtoonify import VToonify

BB = {"D": "dualstylegan", "T": "toonify"}
FP32_TOL = 1e-4    # of max|ref|, the stated fp32 bar


@pytest.fixture
def no_library():
    """CPU tensors of a USER: no emulation bound, nothing loaded."""
    _lib.release_library()
    yield
    _lib.release_library()


# ------------------------------------------------------------------------------ 1. CPU branch
def test_cpu_branch_upfirdn2d_matches_reference_goldens(no_library):
    d, meta = load_golden("op_upfirdn2d.npz")
    for m in meta:
        n = m["name"]
        up = tuple(m["up"]) if isinstance(m["up"], list) else m["up"]
        down = tuple(m["down"]) if isinstance(m["down"], list) else m["down"]
        y = op.upfirdn2d(torch.from_numpy(d[n + "__x"]), torch.from_numpy(d[n + "__k"]), up=up, down=down, pad=tuple(m["pad"]))
        ref = d[n + "__y"]
        assert tuple(y.shape) == ref.shape, n
        if m["integer"]:
            assert np.array_equal(y.numpy(), ref), n
        else:
            assert rel_err(y.numpy(), ref) < 1e-6, n
    assert _lib._lib is None, "the CPU branch must not load a library"


def test_cpu_branch_fused_leaky_relu_matches_reference_goldens(no_library):
    d, meta = load_golden("op_fused_act.npz")
    for m in meta:
        n = m["name"]
        b = torch.from_numpy(d[n + "__b"]) if n + "__b" in d else None
        y = op.fused_leaky_relu(torch.from_numpy(d[n + "__x"]), b, m["slope"], m["scale"])
        assert np.array_equal(y.numpy(), d[n + "__y"]), n
    mod = op.FusedLeakyReLU(5)
    with torch.no_grad():
        mod.bias.copy_(torch.arange(5.0))
    x = torch.randn(2, 5, 3, 3)
    assert torch.equal(mod(x), torch.nn.functional.leaky_relu(x + mod.bias.view(1, 5, 1, 1), 0.2) * 2 ** 0.5)
    assert _lib._lib is None


def test_cpu_branch_is_differentiable_and_conv2d_gradfix_is_torch(no_library):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 9, 7, generator=g, requires_grad=True)
    w = torch.randn(6, 4, 3, 3, generator=g, requires_grad=True)
    y = conv2d_gradfix.conv2d(x, w, padding=1)
    assert torch.equal(y, torch.nn.functional.conv2d(x, w, padding=1))
    wt = torch.randn(4, 3, 3, 3, generator=g)
    assert torch.equal(conv2d_gradfix.conv_transpose2d(x, wt, stride=2), torch.nn.functional.conv_transpose2d(x, wt, stride=2))
    k = synth.fir_kernel_2d(gain=4.0)
    z = op.fused_leaky_relu(op.upfirdn2d(y, k, up=2, pad=(2, 1)), torch.zeros(6))
    gx, gw = torch.autograd.grad(z.pow(2).sum(), [x, w])
    assert gx.shape == x.shape and gw.shape == w.shape and float(gx.abs().sum()) > 0
    # the adjoint identity of upfirdn2d: <U x, v> == <x, U^T v> with U^T from autograd
    a = torch.randn(1, 2, 6, 5, generator=g, requires_grad=True)
    ua = op.upfirdn2d(a, k, up=(2, 1), down=(1, 2), pad=(1, 0, 2, -1))
    v = torch.randn(ua.shape, generator=g)
    ga, = torch.autograd.grad((ua * v).sum(), [a])
    b = torch.randn(a.shape, generator=g)
    assert abs(float((op.upfirdn2d(b, k, up=(2, 1), down=(1, 2), pad=(1, 0, 2, -1)) * v).sum() - (ga * b).sum())) < 1e-3


@pytest.mark.parametrize("tag", ["D", "T"])
def test_module_on_cpu_runs_the_reference_frame(no_library, tag):
    """`VToonify(backbone)` left on the CPU (style_transfer.py:55 `device = "cpu"`): load_state_dict, zplus2wplus, forward."""
    d, _ = load_golden(f"e2e_{tag}.npz")
    m = VToonify(backbone=BB[tag])
    m.load_state_dict(synth.synth_state_dict(load_keys(tag), 0))
    m.eval()
    x, s = torch.from_numpy(d["x"]), torch.from_numpy(d["style"])
    y = m(x, s, d_s=0.5)
    assert tuple(y.shape) == d["y_ds0.5"].shape
    assert rel_err(y.numpy(), d["y_ds0.5"]) < FP32_TOL
    assert rel_err(m.zplus2wplus(torch.from_numpy(d["zplus"])).numpy(), d["wplus"]) < FP32_TOL
    feat, skip = m(x, s, d_s=0.5, return_feat=True)
    assert rel_err(feat.numpy(), d["feat_ds0.5"]) < FP32_TOL and rel_err(skip.numpy(), d["skip_ds0.5"]) < FP32_TOL
    if tag == "D":
        img, masks = m(x, s, d_s=0.5, return_mask=True)
        for i, mk in enumerate(masks):
            assert rel_err(mk.numpy(), d[f"mask{i}_ds0.5"]) < FP32_TOL
        assert rel_err(m(x, s, d_s=0.0).numpy(), d["y_ds0.0"]) < FP32_TOL       # the AdaResBlock early-out
        with pytest.raises(TypeError):
            m(x, s)
    # W-space style and per-sample styles (groups = batch, model.py:273-304)
    assert rel_err(m(x, s[:, 3], d_s=0.5).numpy(), d["y_wspace"]) < FP32_TOL
    y2 = m(torch.from_numpy(d["x2"]), torch.from_numpy(d["style2"]), d_s=0.75)
    assert rel_err(y2.numpy(), d["y2_ds0.75"]) < FP32_TOL
    assert _lib._lib is None, "the CPU path must not load a library"


# ------------------------------------------------------------------------------ 2. the eager graph on the gfx950 library
def _gpu_sd(tag, dev):
    return {k: v.to(dev) for k, v in synth.synth_state_dict(load_keys(tag), 0).items()}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["D", "T"])
def test_reference_eager_graph_on_the_gpu_library_goldens(tag):
    assert torch.cuda.is_available()
    _lib.use_library(_lib.DEFAULT_LIB)
    assert not _lib.is_emulation()
    dev = torch.device("cuda:0")
    d, _ = load_golden(f"e2e_{tag}.npz")
    net = EagerVToonify(_gpu_sd(tag, dev), BB[tag], 256)
    T = lambda a: torch.from_numpy(a).to(dev)
    x, s = T(d["x"]), T(d["style"])
    with torch.no_grad():
        y = net.forward(x, s, 0.5)
        assert rel_err(y.cpu().numpy(), d["y_ds0.5"]) < FP32_TOL
        if tag == "D":
            img, masks = net.forward(x, s, 0.5, return_mask=True)
            for i, mk in enumerate(masks):
                assert rel_err(mk.cpu().numpy(), d[f"mask{i}_ds0.5"]) < FP32_TOL
        y2 = net.forward(T(d["x2"]), T(d["style2"]), 0.75)
        assert rel_err(y2.cpu().numpy(), d["y2_ds0.75"]) < FP32_TOL
        assert rel_err(net.zplus2wplus(T(d["zplus"])).cpu().numpy(), d["wplus"]) < FP32_TOL


@pytest.mark.gpu
def test_reference_eager_graph_on_the_gpu_library_256_vs_oracle():
    """BASELINE configs[1] geometry (22x256x256 -> 3x1024x1024), fp32, VToonify-D: the eager graph on the operator surface
    against the CPU oracle and against the fused executor -- both sides of the boundary give the reference's frame."""
    from oracle import vtoonify_oracle as O   # checker only
    from vtoonify_amd.engine import VToonifyEngine
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(load_keys("D"), 0)
    x, s = synth.synth_frames(1, 256, 256, seed=11), synth.synth_style(seed=12)
    old = O.set_backend("torch")   # full size: minutes with the numpy contractions, seconds with F.conv2d
    try:
        ref = O.vtoonify_forward(synth.to_numpy_sd(sd), x.numpy(), s.numpy(), 0.5, "dualstylegan")
    finally:
        O.set_backend(old)
    sd_dev = {k: v.to(dev) for k, v in sd.items()}
    with torch.no_grad():
        y = EagerVToonify(sd_dev, "dualstylegan", 256).forward(x.to(dev), s.to(dev), 0.5)
    e = rel_err(y.cpu().numpy(), ref)
    print(f"[parity] eager graph on vtoonify_amd.op, 256^2 fp32: max-rel {e:.3e}")
    assert e < FP32_TOL
    z = VToonifyEngine(sd_dev, "dualstylegan", 256, torch.float32, dev).forward(x.to(dev), s.to(dev), 0.5)
    assert rel_err(z.cpu().numpy(), y.cpu().numpy()) < FP32_TOL


# ------------------------------------------------------------------------------ 3. fp64 through the two native operators
def test_fp64_operators(dev):
    d, meta = load_golden("op_upfirdn2d.npz")
    for m in meta:
        n = m["name"]
        up = tuple(m["up"]) if isinstance(m["up"], list) else m["up"]
        down = tuple(m["down"]) if isinstance(m["down"], list) else m["down"]
        x64 = torch.from_numpy(d[n + "__x"]).double().to(dev)
        y = op.upfirdn2d(x64, torch.from_numpy(d[n + "__k"]).to(dev), up=up, down=down, pad=tuple(m["pad"]))
        assert y.dtype == torch.float64 and tuple(y.shape) == d[n + "__y"].shape
        if m["integer"]:
            assert np.array_equal(y.cpu().numpy(), d[n + "__y"].astype(np.float64)), n
        else:
            assert rel_err(y.cpu().numpy(), d[n + "__y"]) < 1e-6, n
    # double arithmetic, not fp32 arithmetic on widened tensors: a sum that fp32 cannot hold
    x = torch.tensor([[[[1.0, 2.0 ** -40]]]], dtype=torch.float64, device=dev)
    k = torch.ones(1, 2, dtype=torch.float64, device=dev)
    assert float(op.upfirdn2d(x, k, pad=(0, 0))[0, 0, 0, 0]) == 1.0 + 2.0 ** -40
    g = torch.Generator().manual_seed(1)
    a = torch.randn(3, 5, 4, 6, generator=g, dtype=torch.float64)
    b = torch.randn(5, generator=g, dtype=torch.float64)
    y = op.fused_leaky_relu(a.to(dev), b.to(dev), 0.2, 2 ** 0.5)
    v = a + b.view(1, 5, 1, 1)
    want = torch.where(v > 0, v, v * float(np.float32(0.2))) * float(np.float32(2 ** 0.5))   # (alpha, scale cross the ABI as float,
    assert y.dtype == torch.float64 and torch.equal(y.cpu(), want)                           #  fused_bias_act.cpp:18-32)
    # second derivative path keeps the dtype
    t = a.to(dev).requires_grad_(True)
    z = op.fused_leaky_relu(op.upfirdn2d(t, torch.ones(2, 2, device=dev) / 4, pad=(1, 0)), None)
    gt, = torch.autograd.grad(z.sum(), [t])
    assert gt.dtype == torch.float64 and gt.shape == t.shape
