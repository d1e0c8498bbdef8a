"""End-to-end parity of the HIP engine (vtoonify_amd.engine / vtoonify_amd.vtoonify) against
(1) golden outputs of the REAL reference (tests/golden/e2e_*.npz, made with its op_cpu
path) and (2) the CPU oracle at full benchmark size on the GPU box.

Stated tolerances (un-clamped output image, relative to the reference's max-abs):
  fp32 mode : max-abs error <= 1e-4 x max|ref|          (measured ~5e-6)
  bf16 mode : PSNR >= 45 dB over the reference's range (SURVEY.md section 8c asks for >= 40 dB; measured on
              MI355X 48-61 dB), 99.9 % of the pixels within 4e-2 x max|ref| (measured <= 3.4e-2), and max-abs error <= 6e-2 x
              max|ref|.  The max-abs of a random-weight network in bf16 is an outlier statistic: 0.8-1.6e-2 at
              the benchmark sizes, but ONE pixel of the 128x128 golden at d_s = 0 sits at 3.0-4.3e-2 and moves by
              +-40 % whenever a layer changes its (fp32) summation order -- so the tightening is carried by the
              PSNR and quantile bars, the max bar stays where round 1 had it.  bf16 has no counterpart in the
              reference; fp32 accumulate everywhere, fp32 statistics / demodulation / RGB skip path)
Every comparison appends (what, dtype, max-rel, PSNR) to gpurun_out/parity_metrics.jsonl when that
directory exists (the GPU box), so the measured margins are on record, not only pass/fail.
"""
import json
import os
import numpy as np
import pytest
import torch

from conftest import load_golden, load_keys, psnr, rel_err
from vtoonify_amd import synth
from vtoonify_amd.engine import VToonifyEngine
from vtoonify_amd.vtoonify import VToonify

FP32_TOL = 1e-4
BF16_TOL, BF16_PSNR, BF16_Q999 = 6e-2, 45.0, 4e-2
_METRICS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
BB = {"D": "dualstylegan", "T": "toonify"}
_cache = {}


def engine(tag, dtype, dev, x3=False):
    key = (tag, dtype, str(dev), x3)
    if key not in _cache:
        _cache.clear()  # one resident engine at a time (D is 666 MB of fp32 weights)
        sd = synth.synth_state_dict(load_keys(tag), 0)
        _cache[key] = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, BB[tag], 256, dtype, dev, x3=x3)
    return _cache[key]


def check(y, ref, dtype, what=""):
    dev_type = y.device.type
    y = y.float().cpu().numpy()
    e = rel_err(y, ref)
    p = psnr(y, ref, float(ref.max() - ref.min()))
    print(f"[parity] {what} {str(dtype).split('.')[-1]} on {dev_type}: max-rel {e:.3e}, PSNR {p:.1f} dB")
    if dev_type == "cuda" and os.path.isdir(_METRICS):
        with open(os.path.join(_METRICS, "parity_metrics.jsonl"), "a") as f:
            f.write(json.dumps({"what": what, "dtype": str(dtype).split(".")[-1], "max_rel": e,
                                "psnr_db": None if np.isinf(p) else p, "shape": list(ref.shape)}) + "\n")
    if dtype == torch.float32:
        assert e < FP32_TOL, f"{what}: {e:.2e}"
    else:
        q = float(np.quantile(np.abs(y.astype(np.float64) - ref), 0.999) / max(np.abs(ref).max(), 1e-30))
        assert e < BF16_TOL and p > BF16_PSNR and q < BF16_Q999, f"{what}: rel {e:.2e}, q99.9 {q:.2e}, psnr {p:.1f} dB"


@pytest.mark.parametrize("tag", ["D", "T"])
def test_golden_fp32(dev, tag):
    d, _ = load_golden(f"e2e_{tag}.npz")
    eng = engine(tag, torch.float32, dev)
    x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
    lean = dev.type == "cpu"   # an emulated fp32 D forward costs ~11 s: the GPU run does the whole list
    for key in [k for k in d if k.startswith("y_ds")]:
        if not (lean and tag == "D"):
            check(eng.forward(x, s, float(key[4:])), d[key], torch.float32, key)
    feat, skip = eng.forward(x, s, 0.5, return_feat=True)
    check(feat, d["feat_ds0.5"], torch.float32, "feat")
    check(skip, d["skip_ds0.5"], torch.float32, "skip")
    if tag == "D":
        img, masks = eng.forward(x, s, 0.5, return_mask=True)
        check(img, d["y_ds0.5"], torch.float32, "image(return_mask)")
        assert len(masks) == 4
        for i, m in enumerate(masks):
            assert tuple(m.shape) == d[f"mask{i}_ds0.5"].shape
            check(m, d[f"mask{i}_ds0.5"], torch.float32, f"mask{i}")
    if lean and tag == "D":
        return  # W-space and per-sample styles: emulated on T (same code path) and in the bf16 test
    check(eng.forward(x, s[:, 3], 0.5), d["y_wspace"], torch.float32, "W-space style")
    # batch 2, non-square 24x40, two different styles (the groups=batch case of model.py:273-304)
    y2 = eng.forward(torch.from_numpy(d["x2"]).to(dev), torch.from_numpy(d["style2"]).to(dev), 0.75)
    check(y2, d["y2_ds0.75"], torch.float32, "per-sample styles")
    check(eng.map_style(torch.from_numpy(d["zplus"]).to(dev)), d["wplus"], torch.float32, "zplus2wplus")


def test_golden_f32x3(dev):
    """The fp32 engine with its convolutions as three bf16 MFMAs per product (VToonifyEngine(x3=True), vt_conv_desc.dtype =
    VT_F32X3) against the REFERENCE's golden frame: the fp32 bar (1e-4 of max|ref|) holds -- measured 2-4e-5 --
    and the result is not the exact-fp32 engine's (the f32x3 instances ran)."""
    d, _ = load_golden("e2e_D.npz")
    eng = engine("D", torch.float32, dev, x3=True)
    x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
    y = eng.forward(x, s, 0.5).clone()
    check(y, d["y_ds0.5"], torch.float32, "y_ds0.5 f32x3")
    feat, skip = eng.forward(x, s, 0.5, return_feat=True)
    check(feat, d["feat_ds0.5"], torch.float32, "feat f32x3")
    e3 = rel_err(y.float().cpu().numpy(), d["y_ds0.5"])
    assert e3 > 0.0
    kinds = {info["kernel"] for _, _, info in eng.frame_ops(eng.plan_for(1, x.shape[2], x.shape[3], True, True)) if isinstance(info, dict)}
    assert any(k.startswith("conv_") for k in kinds)


@pytest.mark.parametrize("tag", ["D", "T"])
def test_golden_bf16(dev, tag):
    d, _ = load_golden(f"e2e_{tag}.npz")
    eng = engine(tag, torch.bfloat16, dev)
    x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
    for key in [k for k in d if k.startswith("y_ds")][:1 if dev.type == "cpu" else None]:
        check(eng.forward(x, s, float(key[4:])), d[key], torch.bfloat16, key)
    y2 = eng.forward(torch.from_numpy(d["x2"]).to(dev), torch.from_numpy(d["style2"]).to(dev), 0.75)
    check(y2, d["y2_ds0.75"], torch.bfloat16, "per-sample styles")


def test_module_dropin_surface(dev):
    """VToonify(backbone).load_state_dict(...) ; model(x, s_w, d_s) as style_transfer.py:62-64,176."""
    d, _ = load_golden("e2e_T.npz")
    m = VToonify(backbone="toonify", compute_dtype=torch.float32)
    m.load_state_dict(synth.synth_state_dict(load_keys("T"), 0))
    m = m.to(dev)
    x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
    y = m(x, s, d_s=0.5)
    check(y, d["y_ds0.5"], torch.float32, "module forward")
    # Toonify ignores the style degree (model/vtoonify.py:238,257): bit-identical outputs
    assert torch.equal(m(x, s, d_s=0.0), y) and torch.equal(m(x, s), y)
    check(m.zplus2wplus(torch.from_numpy(d["zplus"]).to(dev)), d["wplus"], torch.float32, "zplus2wplus")
    assert m.stylegan() is m.generator and m.backbone == "toonify"
    with pytest.raises(Exception, match="multiples of 8"):
        m(torch.zeros(1, 22, 30, 32, device=dev), s)


def test_style_cache_is_keyed_on_the_callers_tensor(dev):
    """with cache_styles=True a CPU / fp64 / fp16 style is converted into a fresh temporary on
    every call; a key made of that temporary's (address, version) can alias a different style.  The cache
    only ever hits on the caller's own device-fp32 tensor object (same version counter)."""
    d, _ = load_golden("e2e_T.npz")
    sd = synth.synth_state_dict(load_keys("T"), 0)
    dt = torch.bfloat16 if dev.type == "cpu" else torch.float32   # emulation: bf16 forwards are 2x cheaper
    nst = 2 if dev.type == "cpu" else 4
    eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, "toonify", 256, dt, dev, cache_styles=True)
    x = torch.from_numpy(d["x"]).to(dev)
    s = torch.from_numpy(d["style"])
    want = {}
    for i in range(nst):   # distinct styles through a converting path (fp64 on the host): never a stale hit
        si = (s + 0.05 * i).double()
        y = eng.forward(x, si, 0.5)
        want[i] = y.clone()
        if i:
            assert not torch.equal(want[i], want[i - 1]), i
    ref = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, "toonify", 256, dt, dev)
    for i in range(nst):
        assert torch.equal(ref.forward(x, (s + 0.05 * i).double(), 0.5), want[i]), i
    # the caller's own tensor: second call hits, an in-place edit (version bump) misses
    own = s.to(dev).clone()
    y0 = eng.forward(x, own, 0.5)
    plan = eng.plan_for(1, x.shape[2], x.shape[3], True, False)
    assert plan.style_ref is own
    assert torch.equal(eng.forward(x, own, 0.5), y0)
    own.add_(0.1)
    y1 = eng.forward(x, own, 0.5)
    assert not torch.equal(y1, y0) and torch.equal(y1, ref.forward(x, own, 0.5))
    # an engine built with device "cuda" / a device string accepts inputs on the indexed device
    assert eng.device == x.device


def test_style_cache_and_determinism(dev):
    lean = dev.type == "cpu"   # emulation: the T backbone (same plan / lane / cache code, 3x cheaper forwards)
    d, _ = load_golden("e2e_T.npz" if lean else "e2e_D.npz")
    eng = engine("T" if lean else "D", torch.bfloat16, dev)
    x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
    y0 = eng.forward(x, s, 0.5)
    if not lean:
        assert torch.equal(eng.forward(x, s, 0.5), y0), "forward must be deterministic (noise is x0)"
    eng.cache_styles = True
    try:
        y1 = eng.forward(x, s, 0.5)   # fills the cache
        y2 = eng.forward(x, s, 0.5)   # style ops skipped
        assert torch.equal(y1, y0) and torch.equal(y2, y0)
        y3 = eng.forward(x, s + 0.1 if lean else s, 1.0)   # key change -> recomputed (T ignores d_s: new style)
        assert not torch.equal(y3, y0)
        # the cache is per plan: lanes that alternate frame by frame each keep theirs
        for lane in ((1, 0) if lean else (1, 0, 1, 0)):
            assert torch.equal(eng.forward(x, s, 0.5, lane=lane), y0), lane
        assert eng._plans[(1, x.shape[2], x.shape[3], True, not lean, 1)].style_key is not None
    finally:
        eng.cache_styles = False
    # shared-style batch == per-frame calls (video path: s_w.repeat(B,1,1), style_transfer.py:176)
    xb = torch.cat([x, x.flip(3)], 0)
    yb = eng.forward(xb, s.repeat(2, 1, 1), 0.5)
    assert torch.equal(yb[:1], y0)
    if not lean:
        assert torch.equal(yb[1:], eng.forward(x.flip(3).contiguous(), s, 0.5))


# ------------------------------------------------------------------ full-size, GPU only
@pytest.mark.gpu
@pytest.mark.parametrize("tag,hw", [("D", (256, 256)), ("T", (144, 256)),
                                    ("T", (256, 256)),        # config 4 (VToonify-T at 1024x1024)
                                    ("D", (384, 384)),        # config 5: 1536x1536
                                    ("D", (360, 400))])       # config 5: the demo's crop, 45x50-pixel trunk (tile edges)
def test_full_size_fp32_vs_oracle(tag, hw):
    """Every BASELINE configuration's frame geometry (configs 1/2: D 256x256; 3: 144x256; 4: T 256x256;
    5: D 384x384 and the non-power-of-two 360x400), fp32 AND bf16 HIP vs the CPU oracle on identical
    seeded weights / inputs."""
    from oracle import vtoonify_oracle as O
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    shapes = load_keys(tag)
    sd = synth.synth_state_dict(shapes, 0)
    x = synth.synth_frames(1, hw[0], hw[1], seed=99)
    s = synth.synth_style(seed=17)
    old = O.set_backend("torch")  # full size: minutes with the numpy contractions, seconds with F.conv2d
    try:
        ref = O.vtoonify_forward(synth.to_numpy_sd(sd), x.numpy(), s.numpy(), 0.5, BB[tag])
    finally:
        O.set_backend(old)
    eng = engine(tag, torch.float32, dev)
    y = eng.forward(x.to(dev), s.to(dev), 0.5)
    assert tuple(y.shape) == (1, 3, 4 * hw[0], 4 * hw[1])
    check(y, ref, torch.float32, f"{tag} {hw}")
    yb = engine(tag, torch.bfloat16, dev).forward(x.to(dev), s.to(dev), 0.5)
    check(yb, ref, torch.bfloat16, f"{tag} {hw} bf16")
    # the fp32 engine on the bf16 matrix cores (three bf16 MFMAs per product): the same 1e-4 bar
    y3 = engine(tag, torch.float32, dev, x3=True).forward(x.to(dev), s.to(dev), 0.5)
    check(y3, ref, torch.float32, f"{tag} {hw} f32x3")
    assert not torch.equal(y3, y)


@pytest.mark.gpu
def test_mid_size_ragged_vs_reference_golden():
    """HIP against the REFERENCE itself (not the oracle) at a mid-size ragged geometry: D, batch 2, 72 x 104, two styles
    (tests/golden/e2e_D_mid.npz).  fp32 exact, fp32 on the bf16 matrix cores (f32x3) and bf16."""
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    d, _ = load_golden("e2e_D_mid.npz")
    sx, s0, s1 = (int(v) for v in d["seeds"])
    h, w = (int(v) for v in d["hw"])
    x = synth.synth_frames(2, h, w, seed=sx).to(dev)
    s = torch.cat([synth.synth_style(seed=s0), synth.synth_style(seed=s1)], 0).to(dev)
    ref = d["y_ds05"]
    check(engine("D", torch.float32, dev).forward(x, s, 0.5), ref, torch.float32, f"D 2x({h},{w}) vs reference golden")
    check(engine("D", torch.float32, dev, x3=True).forward(x, s, 0.5), ref, torch.float32, f"D 2x({h},{w}) f32x3 vs reference golden")
    check(engine("D", torch.bfloat16, dev).forward(x, s, 0.5), ref, torch.bfloat16, f"D 2x({h},{w}) bf16 vs reference golden")


@pytest.mark.gpu
def test_config3_batch4_vs_oracle():
    """BASELINE config 3's per-rank step: D, 4 frames of 22x144x256 per call (the reference's --batch_size 4,
    style_transfer.py:35,176 `s_w.repeat(B,1,1)`), fp32 and bf16 vs the oracle, every frame of the batch."""
    from oracle import vtoonify_oracle as O
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(load_keys("D"), 0)
    x = synth.synth_frames(4, 144, 256, seed=31)
    s = synth.synth_style(seed=17)
    old = O.set_backend("torch")
    try:
        ref = np.concatenate([O.vtoonify_forward(synth.to_numpy_sd(sd), x[i:i + 1].numpy(), s.numpy(), 0.5,
                                                 "dualstylegan") for i in range(4)], 0)
    finally:
        O.set_backend(old)
    for dtype in (torch.float32, torch.bfloat16):
        y = engine("D", dtype, dev).forward(x.to(dev), s.to(dev).repeat(4, 1, 1), 0.5)
        assert tuple(y.shape) == (4, 3, 576, 1024)
        check(y, ref, dtype, "D 4x(144,256)")


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(256, 256), (144, 200)])   # the headline; a size whose 72x100 level has ragged 256-pixel tiles
def test_headline_batch4_vs_oracle(monkeypatch, hw):
    """The bench's headline step: D, 4 frames of 22x256x256 per call, bf16 -- the batch size at which the plans look at the
    batch (256 x 128 patch tiles for the 64^2 / 128^2 convs).  Every frame against the oracle; against the
    same frames one per call (to bf16 rounding by default, bit for bit under VT_BATCH_EXACT=1)."""
    from oracle import vtoonify_oracle as O
    from vtoonify_amd import _lib
    from vtoonify_amd.engine import VToonifyEngine
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(load_keys("D"), 0)
    x = synth.synth_frames(4, hw[0], hw[1], seed=77)
    s = synth.synth_style(seed=17)
    old = O.set_backend("torch")
    try:
        ref = np.concatenate([O.vtoonify_forward(synth.to_numpy_sd(sd), x[i:i + 1].numpy(), s.numpy(), 0.5,
                                                 "dualstylegan") for i in range(4)], 0)
    finally:
        O.set_backend(old)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    xd, sdv = x.to(dev), s.to(dev)
    eng = VToonifyEngine(sdd, "dualstylegan", 256, torch.bfloat16, dev)
    y = eng.forward(xd, sdv.repeat(4, 1, 1), 0.5).clone()
    check(y, ref, torch.bfloat16, f"D 4x{hw} batch-aware plans")
    alone = torch.cat([eng.forward(xd[i:i + 1].contiguous(), sdv, 0.5).clone() for i in range(4)])
    # two bf16 evaluations of the same frame that sum three convs in different orders: they differ like either differs from
    # the fp32 oracle (every activation downstream is rounded to bf16 again), so the same bars apply
    check(y, alone.float().cpu().numpy(), torch.bfloat16, f"D 4x{hw} batch of 4 vs one frame per call")
    monkeypatch.setenv("VT_BATCH_EXACT", "1")
    eng_x = VToonifyEngine(sdd, "dualstylegan", 256, torch.bfloat16, dev)
    assert torch.equal(eng_x.forward(xd, sdv.repeat(4, 1, 1), 0.5), alone)


@pytest.mark.gpu
def test_full_size_properties():
    """Size-independent properties at 1536x1536 output and at the demo's 360x400 crop."""
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    eng = engine("T", torch.bfloat16, dev)
    s = synth.synth_style(seed=17).to(dev)
    for h, w in ((384, 384), (360, 400)):
        x = synth.synth_frames(1, h, w, seed=5).to(dev)
        y = eng.forward(x, s, 0.3)
        assert tuple(y.shape) == (1, 3, 4 * h, 4 * w) and torch.isfinite(y).all()
        assert torch.equal(y, eng.forward(x, s, 0.9)), "Toonify must ignore d_s"
    engd = engine("D", torch.bfloat16, dev)
    x = synth.synth_frames(2, 256, 256, seed=6).to(dev)
    yb = engd.forward(x, s.repeat(2, 1, 1), 0.5)
    assert torch.equal(yb[1:], engd.forward(x[1:].contiguous(), s, 0.5)), "batched == per-frame"
    assert not torch.equal(engd.forward(x[:1].contiguous(), s, 0.0), yb[:1]), "D must depend on d_s"


@pytest.mark.gpu
def test_frames_in_flight_on_separate_lanes():
    """bench.py / video.py keep several frames of a video in flight on one GPU: frame i runs on
    HIP stream i % L with engine lane i % L (own activations, split-K workspace and hipGraph).
    Overlapped frames must be bit-identical to the same frames run one after the other."""
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    eng = engine("D", torch.bfloat16, dev)
    s = synth.synth_style(seed=17).to(dev)
    xs = [synth.synth_frames(1, 256, 256, seed=40 + i).to(dev) for i in range(6)]
    want = [eng.forward(x, s, 0.5, use_graph=True) for x in xs]
    torch.cuda.synchronize()
    L = 3
    streams = [torch.cuda.Stream(dev) for _ in range(L)]
    for rounds in range(2):   # first round builds the lanes' plans and graphs, second one is all replay
        got = []
        for i, x in enumerate(xs):
            with torch.cuda.stream(streams[i % L]):
                got.append(eng.forward(x, s, 0.5, use_graph=True, lane=i % L))
        torch.cuda.synchronize()
        for i in range(len(xs)):
            assert torch.equal(got[i], want[i]), (rounds, i)


@pytest.mark.gpu
def test_two_lanes_in_steady_state_small_frames():
    """Two lanes alternating batches of two small frames for a few hundred steps (what video.py does with depth 2), every
    output compared with the serial result of the same input.  Round 4 found a state of the library in which about one step in
    ten came back with a 16-pixel row of the IMAGE wrong in one channel -- only while the other lane was running, never in the
    six-frame test above (the fused-ToRGB convs on the lean epilogue, conv_igemm.hip::conv_epilogue; tools/flake_lanes.py)."""
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    for tag, d_s, steps in (("D", 0.6, 240), ("T", None, 160)):
        eng = engine(tag, torch.bfloat16, dev)
        s = synth.synth_style(seed=5).to(dev)
        g = torch.Generator().manual_seed(1)
        xs = [torch.randn(2, 22, 64, 96, generator=g).to(dev) for _ in range(6)]
        want = [eng.forward(x, s, d_s, shared_style=True, use_graph=False).clone() for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        pend = [None, None]
        bad = []
        for it in range(steps):
            ln = it % 2
            with torch.cuda.stream(streams[ln]):
                if pend[ln] is not None:
                    y, j = pend[ln]
                    streams[ln].synchronize()
                    if not torch.equal(y, want[j]):
                        bad.append((it, ln, j))
                j = it % len(xs)
                pend[ln] = (eng.forward(xs[j], s, d_s, shared_style=True, use_graph=True, lane=ln + 1).clone(), j)
        torch.cuda.synchronize()
        assert not bad, (tag, len(bad), bad[:5])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,B,H,W,lanes,steps", [("D", 4, 256, 256, 3, 500),    # bench.py's own configuration
                                                   ("D", 2, 64, 96, 3, 800),      # the geometry that showed round 4's defect
                                                   ("D", 3, 72, 104, 3, 500),     # ragged tiles
                                                   ("T", 2, 64, 96, 2, 300)])
def test_lanes_in_steady_state_stress(tag, B, H, W, lanes, steps):
    """2 100 steps with two or three lanes in flight, hipGraph replay, every output `torch.equal` to the serial
    result of the same input.  The library that showed round 4's defect failed this geometry one step in four
    (profiles/r05_torgb_defect.txt: 205 of 800); the cause is a gfx950 instruction form hipcc's SLP vectoriser emits
    (DESIGN.md 4.1n), tests/test_isa_lint.py keeps it out of the library and this test watches what the lint cannot know."""
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    eng = engine(tag, torch.bfloat16, dev)
    d_s = 0.6 if tag == "D" else None
    s = synth.synth_style(seed=5).to(dev)
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(B, 22, H, W, generator=g).to(dev) for _ in range(5)]   # (5 inputs, 2-3 lanes: every lane sees every input)
    want = [eng.forward(x, s, d_s, shared_style=True, use_graph=False).clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
    pend = [None] * lanes
    bad = []
    for it in range(steps):
        ln = it % lanes
        with torch.cuda.stream(streams[ln]):
            if pend[ln] is not None:
                y, j = pend[ln]
                streams[ln].synchronize()
                if not torch.equal(y, want[j]):
                    bad.append((it, ln, j))
            j = it % len(xs)
            pend[ln] = (eng.forward(xs[j], s, d_s, shared_style=True, use_graph=True, lane=ln + 1).clone(), j)
    torch.cuda.synchronize()
    assert not bad, (tag, len(bad), bad[:5])


def test_lanes_are_independent_plans(dev):
    d, _ = load_golden("e2e_T.npz")
    eng = engine("T", torch.float32, dev)
    x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
    y0 = eng.forward(x, s, 0.5)
    y1 = eng.forward(x, s, 0.5, lane=1)
    assert torch.equal(y0, y1)
    p0, p1 = eng.plan_for(1, x.shape[2], x.shape[3], True, False), eng._plans[(1, x.shape[2], x.shape[3], True, False, 1)]
    assert p0 is not p1 and p0.bufs["x_nhwc"].data_ptr() != p1.bufs["x_nhwc"].data_ptr()


def test_tile_hints_override_plans_per_geometry(dev):
    """VToonifyEngine(tile_hints={signature: tile_hint}) (the table tools/plan_sweep.py writes): the hinted
    kernel instance runs, results stay within tolerance, an uncompiled tile is rejected loudly."""
    from vtoonify_amd import _lib
    d, _ = load_golden("e2e_T.npz")
    sd = synth.synth_state_dict(load_keys("T"), 0)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
    eng = VToonifyEngine(sdd, "toonify", 256, torch.float32, dev)
    y0 = eng.forward(x, s, 0.5)
    plan = eng.plan_for(1, x.shape[2], x.shape[3], True, False)
    sigs = {}
    for dsc, info, _, _ in plan.convs:
        sigs.setdefault(info["sig"], info["kernel"])
    # every 3x3 stride-1 geometry onto the 1-D direct-to-LDS / register-staged 64x64 tile with 2 K-slices
    # (":up" = the conv_transpose + blur family, its tile is not a GEMM tile code)
    hints = {sg: 2 * 100000000 + 2 * 1000000 + 64064 for sg in sigs
             if ":k3s1d1p1" in sg and not sg.endswith("nchw") and not sg.endswith(":up")}
    assert len(hints) >= 5
    eng2 = VToonifyEngine(sdd, "toonify", 256, torch.float32, dev, tile_hints=hints)
    y1 = eng2.forward(x, s, 0.5)
    plan2 = eng2.plan_for(1, x.shape[2], x.shape[3], True, False)
    changed = [info for _, info, _, _ in plan2.convs if info["sig"] in hints]
    assert changed and all("64x64" in info["kernel"] and "patch" not in info["kernel"] for info in changed)
    check(y1, d["y_ds0.5"], torch.float32, "hinted plans")
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 1e-5
    bad = dict(hints)
    bad[next(iter(hints))] = 100000000 + 96096     # no such compiled patch tile
    with pytest.raises(_lib.VtError):
        VToonifyEngine(sdd, "toonify", 256, torch.float32, dev, tile_hints=bad).forward(x, s, 0.5)


def test_plan_sweep_candidates_are_wellformed():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "plan_sweep", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "plan_sweep.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    c = ps.candidates("32x32:512->512:k3s1d4p1")
    assert c and all((h // 100000000) in (1, 2) for h in c) and 100000000 + 8 * 1000000 + 128128 in c
    assert all(h % 1000 <= 32 for h in ps.candidates("256x256:136->3:k3s1d1p1:nchw"))
    assert all(h // 100000000 == 2 for h in ps.candidates("64x64:512->512:k3s2d1p1"))


@pytest.mark.parametrize("tag", ["D", "T"])
def test_frame_path_never_reads_the_entries_the_broadcast_skips(dev, tag):
    """frames.broadcast_state_dict(skip_unused=True) sends 104 M of the 166.6 M elements of the D checkpoint: the 4x4 ..
    32x32 generator layers, the noise buffers and the structure transforms of latent rows 0-6 are replaced by zeros on
    every rank (frames.inference_unused).  The frame -- and zplus2wplus -- must not change by a bit."""
    from vtoonify_amd import frames
    sd = synth.synth_state_dict(load_keys(tag), 0)
    skipped = [k for k in sd if frames.inference_unused(k)]
    assert len(skipped) > 50 and not any(k.startswith(("encoder.", "fusion_", "res.")) for k in skipped)
    assert sum(sd[k].numel() for k in skipped) > 0.25 * sum(v.numel() for v in sd.values())
    x = synth.synth_frames(1, 16, 16, seed=4).to(dev)
    s = synth.synth_style(seed=6).to(dev)
    dt = torch.bfloat16 if dev.type == "cpu" else torch.float32
    full = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, BB[tag], 256, dt, dev)
    y0, w0 = full.forward(x, s, 0.5), full.map_style(s)
    del full
    lean = VToonifyEngine({k: (torch.zeros_like(v) if frames.inference_unused(k) else v).to(dev) for k, v in sd.items()},
                          BB[tag], 256, dt, dev)
    assert torch.equal(lean.forward(x, s, 0.5), y0) and torch.equal(lean.map_style(s), w0)


def test_engine_housekeeping_round3(dev, monkeypatch):
    """Round-3 call-path changes: the frame is read in place (fp32 / bf16 / fp16 contiguous) or staged (anything else),
    style rows and d_s are uploaded only when they change, borrow=True hands out the plan's own buffer, the plan cache
    is an LRU (VT_MAX_PLANS), and the drop-in module's precision follows VTOONIFY_AMD_DTYPE / compute_dtype with a
    fingerprint over EVERY parameter."""
    sd = synth.synth_state_dict(load_keys("T"), 0)
    dt = torch.bfloat16 if dev.type == "cpu" else torch.float32
    monkeypatch.setenv("VT_MAX_PLANS", "2")
    eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, "toonify", 256, dt, dev)
    assert eng.max_plans == 2
    x = synth.synth_frames(1, 16, 16, seed=4).to(dev)
    s = synth.synth_style(seed=6).to(dev)
    y0 = eng.forward(x, s, 0.5)
    # non-contiguous view and fp64 input go through the staging copy, fp16 / bf16 are read in place: the layout kernel
    # rounds to the compute dtype either way
    xt = x.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    assert not xt.is_contiguous() and torch.equal(eng.forward(xt, s, 0.5), y0)
    assert torch.equal(eng.forward(x.double(), s, 0.5), y0)
    xh = x.to(torch.bfloat16)
    assert torch.equal(eng.forward(xh, s, 0.5), eng.forward(xh.float(), s, 0.5))
    # the upload of style rows is skipped for the same tensor + version, repeated after an in-place edit
    plan = eng.plan_for(1, 16, 16, True, False)
    assert plan.up_ref is s and torch.equal(eng.forward(x, s, 0.5), y0)
    s2 = s.clone()
    s2.mul_(1.5)
    y2 = eng.forward(x, s2, 0.5)
    assert not torch.equal(y2, y0) and plan.up_ref is s2
    s2.mul_(1.0 / 1.5)                      # same object, new version: must be uploaded again
    assert torch.allclose(eng.forward(x, s2, 0.5).float(), y0.float(), rtol=0, atol=2e-2 * float(y0.float().abs().max()))
    # borrow: the plan's buffer itself, overwritten by the next call on the lane
    yb = eng.forward(x, s, 0.5, borrow=True)
    assert yb.data_ptr() == plan.image.data_ptr() and torch.equal(yb, y0)
    eng.forward(x.flip(3).contiguous(), s, 0.5, borrow=True)
    assert not torch.equal(yb, y0)          # (the view now shows the next frame)
    # LRU: a third shape evicts the least recently used plan
    eng.forward(synth.synth_frames(1, 8, 16, seed=1).to(dev), s, 0.5)
    eng.forward(synth.synth_frames(1, 8, 8, seed=1).to(dev), s, 0.5)
    assert len(eng._plans) == 2 and not any(k[1:3] == (16, 16) for k in eng._plans)
    assert torch.equal(eng.forward(x, s, 0.5), y0)      # rebuilt on demand
    # ---- the module: precision switch + fingerprint
    monkeypatch.delenv("VTOONIFY_AMD_DTYPE", raising=False)
    assert VToonify(backbone="toonify").compute_dtype == torch.float32
    assert VToonify(backbone="toonify").exact_fp32 is False     # fp32 tensors, conv products as three bf16 MFMAs (f32x3)
    monkeypatch.setenv("VTOONIFY_AMD_DTYPE", "fp32_exact")
    assert VToonify(backbone="toonify").compute_dtype == torch.float32 and VToonify(backbone="toonify").exact_fp32 is True
    monkeypatch.setenv("VTOONIFY_AMD_DTYPE", "bf16")
    assert VToonify(backbone="toonify").compute_dtype == torch.bfloat16
    assert VToonify(backbone="toonify", compute_dtype=torch.float32).compute_dtype == torch.float32
    monkeypatch.setenv("VTOONIFY_AMD_DTYPE", "fp8")
    with pytest.raises(ValueError):
        VToonify(backbone="toonify")
    monkeypatch.setenv("VTOONIFY_AMD_DTYPE", "bf16")
    m = VToonify(backbone="toonify")
    m.load_state_dict(sd)
    m = m.to(dev)
    e0 = m.engine()
    assert m.engine() is e0
    with torch.no_grad():
        m.fusion_skip[0].bias.add_(0.25)    # an in-place edit of a parameter that is neither first, middle nor last
    assert m.engine() is not e0


def test_fusion_gate_in_the_loader_is_bit_identical(dev, monkeypatch):
    """The Fusion gate's mask conv with cat[f_G, |f_G - f_E|] + AdaIN affine formed in its loader (vt_conv_desc.in_absdiff,
    the default from the H/4 level up) against vt_affine_apply + plain conv: the same frames, bit for bit."""
    sd = synth.synth_state_dict(load_keys("D"), 0)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    x = synth.synth_frames(2, 16, 24, seed=3).to(dev)
    s = synth.synth_style(seed=17).to(dev)
    for dtype in ((torch.bfloat16,) if dev.type != "cuda" else (torch.float32, torch.bfloat16)):   # (CPU emulation: one dtype)
        outs = []
        for mode in ("0", "2"):                      # never / at every level (the default skips the H/8 level)
            monkeypatch.setenv("VT_GATE_LOADER", mode)
            eng = VToonifyEngine(sdd, "dualstylegan", 256, dtype, dev)
            outs.append(eng.forward(x, s.repeat(2, 1, 1), 0.5).clone())
            n_aff = sum(1 for op in eng._plans[next(iter(eng._plans))].gen_ops if op[2].get("kernel") == "affine_apply")
            assert n_aff == (4 if mode == "0" else 0)
        assert torch.equal(outs[0], outs[1]), dtype


def test_style_gate_is_transparent(dev):
    """VToonifyEngine(style_gate=True) (what the drop-in module uses): the style path is skipped ON THE DEVICE when a call's
    W+ rows and d_s equal the ones its products were computed from -- the video loop's `s_w.repeat(B,1,1)` is a new tensor
    with the same content on every call (style_transfer.py:176).  Outputs are bit-identical to recomputing; a changed
    row, a changed d_s (D) and a return to an earlier style all recompute."""
    lean = dev.type == "cpu"
    tag = "T" if lean else "D"
    sd = {k: v.to(dev) for k, v in synth.synth_state_dict(load_keys(tag), 0).items()}
    dt = torch.bfloat16
    ref = VToonifyEngine(sd, BB[tag], 256, dt, dev)
    eng = VToonifyEngine(sd, BB[tag], 256, dt, dev, style_gate=True)
    x = synth.synth_frames(2, 16, 16, seed=4).to(dev)
    s = synth.synth_style(seed=6).to(dev)
    y0 = ref.forward(x, s.repeat(2, 1, 1), 0.5)
    flag = lambda: eng.plan_for(2, 16, 16, True, not lean).bufs["style_gate"].cpu().tolist()
    for i in range(3):
        assert torch.equal(eng.forward(x, s.repeat(2, 1, 1), 0.5), y0), i     # a NEW tensor object every call
        # computed once, then skipped (on a GPU the first call is a warm-up launch + the capture's replay: the replay
        # already finds the rows unchanged)
        assert flag() == [1 if (i == 0 and lean) else 0, 0], (i, flag())
    s2 = s + 0.125
    y2 = ref.forward(x, s2.repeat(2, 1, 1), 0.5)
    assert not torch.equal(y2, y0)
    assert torch.equal(eng.forward(x, s2.repeat(2, 1, 1), 0.5), y2) and flag()[0] == 1
    assert torch.equal(eng.forward(x, s2.repeat(2, 1, 1), 0.5), y2) and flag()[0] == 0
    assert torch.equal(eng.forward(x, s.repeat(2, 1, 1), 0.5), y0) and flag()[0] == 1
    if not lean:   # the style degree is part of the gate (VToonify-D; T ignores it)
        y3 = ref.forward(x, s.repeat(2, 1, 1), 0.75)
        assert not torch.equal(y3, y0)
        assert torch.equal(eng.forward(x, s.repeat(2, 1, 1), 0.75), y3) and flag()[0] == 1
        assert torch.equal(eng.forward(x, s.repeat(2, 1, 1), 0.75), y3) and flag()[0] == 0
    # W-space styles (model/vtoonify.py:212-215) go through the same gate
    yw = ref.forward(x, s[:, 3].repeat(2, 1), 0.5)
    assert torch.equal(eng.forward(x, s[:, 3].repeat(2, 1), 0.5), yw)
    assert torch.equal(eng.forward(x, s[:, 3].repeat(2, 1), 0.5), yw) and flag()[0] == 0
