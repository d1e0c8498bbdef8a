"""RAFT optical-flow network (vtoonify_amd/raft.py) against tensors computed by the REAL reference
(model.raft.core.raft.RAFT on the CPU with a synthetic state_dict of its schema, tests/golden/make_golden_raft_net.py):
state_dict schema, module surface, every stage of the first refinement iteration, the flow after 2-3 iterations.

Tolerance, fp32 relative to max|ref|: 1e-4 on the flows (measured ~6e-6 after three iterations; only summation order
differs: MFMA tiles, folded BatchNorm, the memory-efficient correlation lookup instead of the all-pairs volume)."""
import argparse

import numpy as np
import pytest
import torch

from conftest import load_golden, load_keys, rel_err
from vtoonify_amd import _lib, kernels as K, synth
from vtoonify_amd.raft import RAFT, RaftEngine, raft_schema

TOL = 1e-4


def test_state_dict_schema_matches_reference():
    shapes = load_keys("raft")
    assert {k: tuple(v) for k, v in raft_schema().items()} == shapes and len(shapes) == 179
    m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert own == shapes
    assert list(m.state_dict().keys())[:3] == ["fnet.conv1.weight", "fnet.conv1.bias", "fnet.layer1.0.conv1.weight"]
    with pytest.raises(_lib.VtError):
        RAFT(argparse.Namespace(small=True))


def test_oracle_pinned_to_reference_raft():
    """oracle/raft_oracle.py:raft_forward against the reference's RAFT class on a crop small enough for the pure-numpy
    correlation loop (the reference needs >= 128x128: four pyramid levels of at least 2x2)."""
    from oracle import raft_oracle as R
    d, _ = load_golden("raft_net.npz")
    sd = synth.to_numpy_sd(synth.synth_state_dict(load_keys("raft"), 0))
    lo, up = R.raft_forward(sd, d["b__image1"][:1].astype(np.float32), d["b__image2"][:1].astype(np.float32),
                            int(d["b__cfg"][0]))
    assert rel_err(lo, d["b__flow_low"][:1]) < TOL and rel_err(up, d["b__flow_up"][:1]) < TOL


def test_first_iteration_stage_by_stage(dev):
    d, _ = load_golden("raft_net.npz")
    eng = RaftEngine(synth.synth_state_dict(load_keys("raft"), 0), torch.float32, dev)
    eng.keep_taps = True
    im1, im2 = (torch.from_numpy(d[k].astype(np.float32)).to(dev) for k in ("a__image1", "a__image2"))
    lo, ups = eng.forward(im1, im2, iters=int(d["a__cfg"][0]))
    T = eng.taps
    nchw = lambda t: t.permute(0, 3, 1, 2).cpu().numpy()
    assert rel_err(nchw(T["fmap1"])[:, ::4], d["a__fmap1"]) < TOL          # feature encoder (InstanceNorm)
    g = d["a__cnet"]                                                       # context encoder (folded BatchNorm)
    want = np.concatenate([np.tanh(g[:, :32]), np.maximum(g[:, 32:], 0)], 1)   # [tanh(net) | relu(inp)], every 4th channel
    assert rel_err(nchw(T["cnet"])[:, ::4], want) < TOL
    assert rel_err(T["corr1"].cpu().numpy()[:, ::9], d["a__corr1"]) < TOL  # 4-level lookup at zero flow
    assert rel_err(nchw(T["net1"])[:, ::2], d["a__net1"]) < 5e-4           # SepConvGRU output in [-1, 1]
    assert rel_err(T["delta1"].cpu().numpy(), d["a__delta1"]) < TOL        # flow head
    assert rel_err(lo.cpu().numpy(), d["a__flow_low"]) < TOL
    assert len(ups) == 1 and rel_err(ups[0].cpu().numpy(), d["a__flow_up"]) < TOL   # convex up-sampling


def test_module_surface_like_smooth_parsing_map(dev):
    """RAFT(args).load_state_dict(...); model(image1, image2, iters=..., test_mode=True) (smooth_parsing_map.py:97-102,154)."""
    d, _ = load_golden("raft_net.npz")
    m = RAFT(argparse.Namespace(model="raft-things.pth", small=False, mixed_precision=False, alternate_corr=False))
    m.load_state_dict(synth.synth_state_dict(load_keys("raft"), 0))
    m = m.to(dev).eval()
    im1, im2 = (torch.from_numpy(d[k].astype(np.float32)).to(dev) for k in ("b__image1", "b__image2"))
    iters = int(d["b__cfg"][0])
    flow_low, flow_up = m(im1, im2, iters=iters, test_mode=True)
    assert tuple(flow_up.shape) == d["b__flow_up"].shape
    assert rel_err(flow_low.cpu().numpy(), d["b__flow_low"]) < TOL
    assert rel_err(flow_up.cpu().numpy(), d["b__flow_up"]) < TOL
    if dev.type == "cuda":     # training-mode return value: one up-sampled flow per iteration (raft.py:137-144)
        preds = m(im1, im2, iters=iters)
        assert len(preds) == iters and torch.equal(preds[-1], flow_up)
    with pytest.raises(_lib.VtError, match="multiples of 8"):
        m(im1[:, :, :-3], im2[:, :, :-3])


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference"), reason="reference checkout not mounted")
def test_reference_construction_lines_run_on_the_mirror(tmp_path):
    """smooth_parsing_map.py:91-102 executed verbatim (read from the reference at run time) with
    `from model.raft.core.raft import RAFT` resolved by the mirror package: argparse -> RAFT(args) ->
    nn.DataParallel -> load_state_dict of a `module.`-prefixed checkpoint -> .module -> .to(device) -> .eval()."""
    import os
    import sys
    import textwrap
    from emu import build_emu
    _lib.use_library(build_emu.build())
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert sys.path[0] == repo or repo in sys.path
    from model.raft.core.raft import RAFT as MirrorRAFT
    assert MirrorRAFT is RAFT
    sd = synth.synth_state_dict(load_keys("raft"), 0)
    ck = tmp_path / "raft-things.pth"
    torch.save({"module." + k: v for k, v in sd.items()}, ck)
    src = "/root/reference/smooth_parsing_map.py"
    with open(src) as f:
        code = textwrap.dedent("".join(f.readlines()[90:102]))
    ns = {"torch": torch, "argparse": argparse, "RAFT": MirrorRAFT, "device": "cpu",
          "args": argparse.Namespace(raft_path=str(ck))}
    exec(compile(code, src, "exec"), ns)
    m = ns["raft_model"]
    assert isinstance(m, RAFT) and not m.training
    d, _ = load_golden("raft_net.npz")
    im1, im2 = (torch.from_numpy(d[k][:1].astype(np.float32)) for k in ("b__image1", "b__image2"))
    flow_low, flow_up = m(im1, im2, iters=int(d["b__cfg"][0]), test_mode=True)
    assert rel_err(flow_up.numpy(), d["b__flow_up"][:1]) < TOL


def test_asymmetric_padding_convs(dev):
    """vt_conv_desc.pad_w_p1: the (1,5) pad (0,2) and (5,1) pad (2,0) convs of SepConvGRU (update.py:37-42), two
    sources, sigmoid / tanh epilogues."""
    g = np.random.default_rng(5)
    x0 = g.standard_normal((2, 16, 7, 9)).astype(np.float32)
    x1 = g.standard_normal((2, 24, 7, 9)).astype(np.float32)
    a, b = K.nchw_to_nhwc(torch.from_numpy(x0).to(dev), torch.float32), K.nchw_to_nhwc(torch.from_numpy(x1).to(dev), torch.float32)
    for kh, kw, pad, pad_w, act, fn in ((1, 5, 0, 2, _lib.ACT_SIGMOID, torch.sigmoid), (5, 1, 2, 0, _lib.ACT_TANH, torch.tanh)):
        w = (g.standard_normal((12, 40, kh, kw)) / 8).astype(np.float32)
        bias = g.standard_normal(12).astype(np.float32)
        wp = K.pack_conv_weight(torch.from_numpy(w).to(dev), out_dtype=torch.float32)
        out = torch.zeros((2, 7, 9, 16), device=dev)
        K.conv2d(src0=a, c0=16, ld0=16, src1=b, c1=24, ld1=24, n=2, h=7, w=9, out_h=7, out_w=9, weight=wp, cout=12, kh=kh,
                 kw=kw, pad=pad, pad_w=pad_w, bias=torch.from_numpy(bias).to(dev), act=act, out=out, ld_out=16, dtype=K.VT_F32)
        ref = fn(torch.nn.functional.conv2d(torch.from_numpy(np.concatenate([x0, x1], 1)), torch.from_numpy(w),
                                            torch.from_numpy(bias), padding=(pad, pad_w)))
        assert rel_err(out.cpu().permute(0, 3, 1, 2).numpy()[:, :12], ref.numpy()) < 1e-5, (kh, kw)


def test_glue_kernels(dev):
    """vt_eltwise2 / vt_gru_blend / vt_coords_from_flow / vt_convex_upsample against their torch formulas
    (update.py:48-55, raft.py:58-84)."""
    import ctypes as C
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = lambda t: K._stream(t)
    a, b = torch.randn(30, 24, generator=g).to(dev), torch.randn(30, 40, generator=g).to(dev)     # b: wider rows
    for op, fn in ((0, lambda x, y: x * y), (1, lambda x, y: x + y), (2, lambda x, y: torch.relu(x + y))):
        out = torch.zeros(30, 32, device=dev)
        _lib.check(lib.vt_eltwise2(p(out), 32, p(a), 24, p(b), 40, 30, 24, op, K.VT_F32, st(a)), "eltwise2")
        assert torch.allclose(out[:, :24].cpu(), fn(a.cpu(), b[:, :24].cpu()), atol=1e-6) and float(out[:, 24:].abs().max()) == 0
    h = torch.randn(30, 40, generator=g).to(dev)
    z, q = torch.rand(30, 24, generator=g).to(dev), torch.randn(30, 24, generator=g).to(dev)
    want = (1 - z.cpu()) * h[:, :24].cpu() + z.cpu() * q.cpu()
    tail = h[:, 24:].clone()
    _lib.check(lib.vt_gru_blend(p(h), 40, p(z), p(q), 30, 24, K.VT_F32, st(h)), "gru_blend")
    assert torch.allclose(h[:, :24].cpu(), want, atol=1e-6) and torch.equal(h[:, 24:], tail)
    flow = (torch.randn(2, 2, 5, 7, generator=g) * 3).to(dev)
    coords = torch.zeros(2, 1, 5, 7, 2, device=dev)
    _lib.check(lib.vt_coords_from_flow(p(coords), p(flow), 2, 5, 7, st(flow)), "coords")
    ys, xs = torch.meshgrid(torch.arange(5.0), torch.arange(7.0), indexing="ij")
    want = torch.stack([xs, ys], -1)[None, None] + flow.cpu().permute(0, 2, 3, 1)[:, None]
    assert torch.allclose(coords.cpu(), want, atol=1e-6)
    mask = torch.randn(2, 576, 5, 7, generator=g).to(dev)
    up = torch.zeros(2, 2, 40, 56, device=dev)
    _lib.check(lib.vt_convex_upsample(p(up), p(flow), p(mask), 2, 5, 7, st(flow)), "convex")
    m = torch.softmax(mask.cpu().view(2, 1, 9, 8, 8, 5, 7), dim=2)                       # raft.py:75-84
    uf = torch.nn.functional.unfold(8 * flow.cpu(), [3, 3], padding=1).view(2, 2, 9, 1, 1, 5, 7)
    want = torch.sum(m * uf, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(2, 2, 40, 56)
    assert rel_err(up.cpu().numpy(), want.numpy()) < 1e-5


@pytest.mark.gpu
def test_video_size_pair_is_finite_and_deterministic():
    """512x512 frames (smooth_parsing_map.py:127), 20 iterations: finite, deterministic, and the flow feeds the
    parsing-map fusion (vtoonify_amd.smooth) end to end."""
    from vtoonify_amd import smooth
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
    m.load_state_dict(synth.synth_state_dict(load_keys("raft"), 0))
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(9)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, 520, 520, generator=g), 5, stride=1, padding=2)
    w = 1
    Is = torch.stack([base[0, :, 4 + k:516 + k, 4:516] for k in range(3)]).to(dev) * 2 - 1       # a slowly moving clip
    Ps = (torch.randn(3, 19, 512, 512, generator=g) * 4).to(dev)
    flow_fn = smooth.raft_flow_fn(m, iters=20)      # smooth_parsing_map.py:154
    f1 = flow_fn(Is[1:2].repeat(3, 1, 1, 1), Is)
    assert tuple(f1.shape) == (3, 2, 512, 512) and bool(torch.isfinite(f1).all())
    assert torch.equal(f1, flow_fn(Is[1:2].repeat(3, 1, 1, 1), Is))
    # the opt-in hipGraph replay (VT_RAFT_GRAPH=1, captured on the second call of a shape) issues the same launches
    import os
    os.environ["VT_RAFT_GRAPH"] = "1"
    try:
        for _ in range(3):
            assert torch.equal(f1, flow_fn(Is[1:2].repeat(3, 1, 1, 1), Is))
        assert any(isinstance(v, tuple) for v in m.engine()._graphs.values())
    finally:
        os.environ["VT_RAFT_GRAPH"] = "0"
    y = smooth.smooth_parsing_maps(Is, Ps, flow_fn, w)
    assert tuple(y.shape) == (3, 19, 256, 256) and bool(torch.isfinite(y).all())
    # ---- at the WORKING size and iteration count against the CPU oracle (a restatement of raft.py:86-144 pinned to
    # reference-made goldens at the small size): 20 GRU iterations of exact-fp32 MFMA against torch-CPU convolutions.
    # Stated drift bar: 2e-3 of max|flow| (the flows differ by summation order only; 3 iterations measure 6e-6).
    import json
    import os
    from oracle import raft_oracle as R, vtoonify_oracle as O
    O.set_backend("torch")
    i1 = ((Is[1:2] + 1) * 255.0 / 2)
    i2 = ((Is[2:3] + 1) * 255.0 / 2)
    lo, up = m(i1, i2, iters=20, test_mode=True)
    wl, wu = R.raft_forward(synth.to_numpy_sd(synth.synth_state_dict(load_keys("raft"), 0)), i1.cpu().numpy(),
                            i2.cpu().numpy(), iters=20)
    el, eu = rel_err(lo.cpu().numpy(), wl), rel_err(up.cpu().numpy(), wu)
    print(f"[parity] RAFT 512x512 x 20 iterations: flow_low {el:.2e}, flow_up {eu:.2e}, max|flow| {np.abs(wu).max():.2f}")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_metrics.jsonl"), "a") as f:
            f.write(json.dumps({"what": "RAFT 512x512 x 20 iterations flow_up vs oracle", "dtype": "float32",
                                "max_rel": eu, "psnr_db": None, "shape": list(wu.shape)}) + "\n")
    assert el < 2e-3 and eu < 2e-3, (el, eu)
