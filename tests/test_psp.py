"""pSp style encoder (vtoonify_amd/psp.py) against golden outputs of the REAL
reference (tests/golden/psp.npz) and, at the full 3x256x256 size on the GPU, the CPU oracle.

Tolerances (W+ codes, relative to max|ref|): fp32 1e-4; bf16 5e-2 (24 residual units + FPN + heads
in bf16 with fp32 accumulation; the reference has no bf16 path)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, load_keys, rel_err
from vtoonify_amd import synth
from vtoonify_amd.psp import GradualStyleEncoder, PspEngine

TOL = {torch.float32: 1e-4, torch.bfloat16: 5e-2}


def test_state_dict_schema_matches_reference():
    shapes = load_keys("psp")
    m = GradualStyleEncoder(50, "ir_se")
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert own == shapes and len(own) == 621


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_golden_small(dev, dtype):
    d, _ = load_golden("psp.npz")
    sd = synth.synth_state_dict(load_keys("psp"), 0)
    eng = PspEngine({k: v.to(dev) for k, v in sd.items()}, 18, dtype, dev)
    x = torch.from_numpy(d["s32__x"]).to(dev)
    y, taps = eng.forward(x, taps=True)
    assert tuple(y.shape) == (2, 18, 512)
    assert rel_err(taps[6].cpu().numpy(), d["s32__c1"]) < TOL[dtype]
    assert rel_err(taps[23].cpu().numpy(), d["s32__c3"]) < TOL[dtype]
    assert rel_err(y.cpu().numpy(), d["s32__y"]) < TOL[dtype]
    if dev.type == "cpu" and dtype == torch.float32:
        return   # emulation: the batch-independence forward runs in the (2x cheaper) bf16 case and on the GPU
    # batch independence: frame 1 alone
    y1 = eng.forward(x[1:].contiguous())
    assert rel_err(y1.cpu().numpy(), d["s32__y"][1:]) < TOL[dtype]


def test_module_surface(dev):
    """load_state_dict + call like util.load_psp_standalone (util.py:143-161), incl. latent_avg."""
    d, _ = load_golden("psp.npz")
    dt = torch.bfloat16 if dev.type == "cpu" else torch.float32   # emulation: the cheaper arithmetic
    m = GradualStyleEncoder(50, "ir_se", compute_dtype=dt)
    m.load_state_dict(synth.synth_state_dict(load_keys("psp"), 0))
    m = m.to(dev)
    x = torch.from_numpy(d["s32__x"][:1]).to(dev)
    y = m(x)
    assert rel_err(y.cpu().numpy(), d["s32__y"][:1]) < TOL[dt]
    m.latent_avg = torch.full((18, 512), 0.5, device=dev)
    assert torch.allclose(m(x), y + 0.5)


@pytest.mark.gpu
def test_full_size_vs_oracle():
    from oracle import psp_oracle as P, vtoonify_oracle as O
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(load_keys("psp"), 0)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    old = O.set_backend("torch")
    try:
        ref = P.gradual_style_encoder(synth.to_numpy_sd(sd), x.numpy())
    finally:
        O.set_backend(old)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    for dtype in (torch.float32, torch.bfloat16):
        eng = PspEngine(sdd, 18, dtype, dev)
        y = eng.forward(x.to(dev))
        assert rel_err(y.cpu().numpy(), ref) < TOL[dtype], dtype
        yg = eng.forward(x.to(dev), use_graph=True)
        assert torch.equal(yg, eng.forward(x.to(dev), use_graph=True))
        assert rel_err(yg.cpu().numpy(), ref) < TOL[dtype]
