"""BiSeNet face parsing (vtoonify_amd/bisenet.py) against golden outputs of the
REAL reference (tests/golden/bisenet.npz, made by tests/golden/make_golden_bisenet.py) and, at the
video loop's 512x512 network size on the GPU, the CPU oracle.

Tolerances (class maps / feature taps, relative to max|ref|): fp32 1e-4; bf16 4e-2 (ResNet18 + ARM +
FFM in bf16 with fp32 accumulation; the reference has no bf16 path)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, load_keys, rel_err
from vtoonify_amd import synth
from vtoonify_amd.bisenet import BiSeNet, BiSeNetEngine

TOL = {torch.float32: 1e-4, torch.bfloat16: 4e-2}


def test_state_dict_schema_matches_reference():
    shapes = load_keys("bisenet")
    own = {k: tuple(v.shape) for k, v in BiSeNet(19).state_dict().items()}
    assert own == shapes and len(own) == 191


def test_oracle_pinned_to_reference():
    from oracle import bisenet_oracle as B, vtoonify_oracle as O
    d, _ = load_golden("bisenet.npz")
    sd = synth.to_numpy_sd(synth.synth_state_dict(load_keys("bisenet"), 0))
    old = O.set_backend("torch")
    try:
        for name in ("s64", "s96x64"):
            outs, taps = B.bisenet_forward(sd, d[name + "__x"], return_taps=True)
            for t, k in zip(taps, ("__res8", "__cp8", "__cp16")):
                assert rel_err(t, d[name + k]) < 2e-5, (name, k)
            assert rel_err(outs[0], d[name + "__y"]) < 2e-5, name
        assert rel_err(outs[0], d["s96x64__y"]) < 2e-5
        outs = B.bisenet_forward(sd, d["s64__x"])
        assert rel_err(outs[1], d["s64__y16"]) < 2e-5 and rel_err(outs[2], d["s64__y32"]) < 2e-5
        for name in ("frame32", "frame40x24"):
            assert rel_err(B.parsing_maps(sd, d[name + "__x"]), d[name + "__xp"]) < 2e-5, name
    finally:
        O.set_backend(old)
    # numpy contractions (no torch in the arithmetic) on the smallest case
    assert rel_err(B.parsing_maps(sd, d["frame32__x"]), d["frame32__xp"]) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_golden(dev, dtype):
    d, _ = load_golden("bisenet.npz")
    sd = synth.synth_state_dict(load_keys("bisenet"), 0)
    eng = BiSeNetEngine({k: v.to(dev) for k, v in sd.items()}, 19, dtype, dev)
    tol = TOL[dtype]
    x = torch.from_numpy(d["s64__x"]).to(dev)
    (y, y16, y32), taps = eng.forward(x, taps=True)
    for k in ("res8", "cp8", "cp16"):
        assert rel_err(taps[k].cpu().numpy(), d["s64__" + k]) < tol, k
    assert rel_err(y.cpu().numpy(), d["s64__y"]) < tol
    assert rel_err(y16.cpu().numpy(), d["s64__y16"]) < tol and rel_err(y32.cpu().numpy(), d["s64__y32"]) < tol
    # batch 2, non-square; batch independence
    x2 = torch.from_numpy(d["s96x64__x"]).to(dev)
    y2 = eng.forward(x2)[0]
    assert rel_err(y2.cpu().numpy(), d["s96x64__y"]) < tol
    y2b = eng.forward(x2[1:].contiguous())[0]
    assert rel_err(y2b.cpu().numpy(), d["s96x64__y"][1:]) < tol
    # the video loop's pre/post-processing fused around the net (style_transfer.py:171-172)
    for name in ("frame32", "frame40x24"):
        xp = eng.parsing_maps(torch.from_numpy(d[name + "__x"]).to(dev))
        assert tuple(xp.shape) == d[name + "__xp"].shape
        assert rel_err(xp.cpu().numpy(), d[name + "__xp"]) < tol, name


def test_module_surface(dev):
    """parsingpredictor = BiSeNet(n_classes=19); load_state_dict; .to(device).eval(); net(x)[0]
    (style_transfer.py:66-68, 171)."""
    d, _ = load_golden("bisenet.npz")
    m = BiSeNet(n_classes=19, compute_dtype=torch.float32)
    m.load_state_dict(synth.synth_state_dict(load_keys("bisenet"), 0))
    m = m.to(dev).eval()
    y = m(torch.from_numpy(d["s64__x"]).to(dev))[0]
    assert rel_err(y.cpu().numpy(), d["s64__y"]) < 1e-4
    with pytest.raises(Exception, match="3,H,W"):
        m(torch.zeros(1, 4, 32, 32, device=dev))


@pytest.mark.gpu
def test_full_size_vs_oracle():
    """The video loop's size: 256x256 frames -> the net runs at 512x512 -> (19,256,256) maps."""
    from oracle import bisenet_oracle as B, vtoonify_oracle as O
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(load_keys("bisenet"), 0)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    old = O.set_backend("torch")
    try:
        ref = B.parsing_maps(synth.to_numpy_sd(sd), x.numpy())
    finally:
        O.set_backend(old)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    for dtype in (torch.float32, torch.bfloat16):
        eng = BiSeNetEngine(sdd, 19, dtype, dev)
        y = eng.parsing_maps(x.to(dev))
        assert tuple(y.shape) == (2, 19, 256, 256)
        assert rel_err(y.cpu().numpy(), ref) < TOL[dtype], dtype
        yg = eng.parsing_maps(x.to(dev), use_graph=True)
        assert torch.equal(yg, eng.parsing_maps(x.to(dev), use_graph=True))
        assert torch.equal(yg, y), "graph replay == eager launches"
        y1 = eng.parsing_maps(x[1:].to(dev).contiguous())
        assert rel_err(y1.cpu().numpy(), ref[1:]) < TOL[dtype]
