"""Frame pack / unpack kernels and the video driver.

uint8 / index work: BIT-exact.  The fp32 network input is bit-exact too (same op sequence as
ToTensor + Normalize, one rounding per op)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden, load_keys
sys.path.insert(0, os.path.join(REPO, "oracle"))
import frames_oracle as FO  # noqa: E402
from vtoonify_amd import synth, video  # noqa: E402
from vtoonify_amd.engine import VToonifyEngine  # noqa: E402


def test_oracle_reproduces_golden():
    d, _ = load_golden("frames_io.npz")
    x = FO.pack_inputs(d["frames_bgr"], d["parsing"])
    assert x.dtype == np.float32 and np.array_equal(x, d["inputs"])
    out = np.stack([FO.tensor2cv2(y) for y in d["y"]], 0)
    assert out.dtype == np.uint8 and np.array_equal(out, d["out_bgr"])
    # every uint8 level maps to the same float the torch formulas give, and back
    lv = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    t = FO.to_tensor_normalize(lv)
    assert t.min() == -1.0 and t.max() == 1.0
    back = FO.tensor2cv2(t)[..., ::-1].astype(np.int32)
    # astype(uint8) truncates (util.py:191), so a level may come back one lower -- never higher
    assert ((lv.astype(np.int32) - back) >= 0).all() and ((lv.astype(np.int32) - back) <= 1).all()


def test_frame_pack_unpack_golden(dev):
    d, _ = load_golden("frames_io.npz")
    f, p = torch.from_numpy(d["frames_bgr"]).to(dev), torch.from_numpy(d["parsing"]).to(dev)
    x = video.frame_pack(f, p, bgr=True)
    assert np.array_equal(x.cpu().numpy(), d["inputs"])
    out = video.frame_unpack(torch.from_numpy(d["y"]).to(dev), bgr=True)
    assert np.array_equal(out.cpu().numpy(), d["out_bgr"])


@pytest.mark.parametrize("shape", [(1, 5, 7), (3, 8, 8), (2, 9, 12)])   # hw % 4 != 0 -> scalar path
@pytest.mark.parametrize("bgr", [True, False])
def test_frame_pack_unpack_vs_oracle(dev, shape, bgr):
    n, h, w = shape
    g = np.random.default_rng(n * 100 + h)
    frames = g.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    parsing = (g.standard_normal((n, 4, h, w)) * 5).astype(np.float32)
    ref = FO.pack_inputs(frames if bgr else frames[..., ::-1], parsing)
    x = video.frame_pack(torch.from_numpy(frames).to(dev), torch.from_numpy(parsing).to(dev), bgr=bgr)
    assert np.array_equal(x.cpu().numpy(), ref)
    x3 = video.frame_pack(torch.from_numpy(frames).to(dev), None, bgr=bgr)       # no parsing map
    assert np.array_equal(x3.cpu().numpy(), ref[:, :3])
    y = (g.standard_normal((n, 3, h, w)) * 0.9).astype(np.float32)
    y.reshape(-1)[:4] = [-7.0, 7.0, -1.0, 1.0]
    ref_o = np.stack([FO.tensor2cv2(v) for v in y], 0)
    if not bgr:
        ref_o = ref_o[..., ::-1]
    out = video.frame_unpack(torch.from_numpy(y).to(dev), bgr=bgr)
    assert np.array_equal(out.cpu().numpy(), ref_o)


def test_frame_pack_rejects_bad_arguments(dev):
    from vtoonify_amd import _lib
    f = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.VtError):
        video.frame_pack(f.float())
    with pytest.raises(_lib.VtError):
        video.frame_pack(f, torch.zeros((1, 19, 4, 4), device=dev))
    with pytest.raises(_lib.VtError):
        video.frame_unpack(torch.zeros((1, 4, 8, 8), device=dev))


def _video_case(dev, n_frames, H, W, batch, depth, dtype):
    g = np.random.default_rng(3)
    frames = g.integers(0, 256, (n_frames, H, W, 3), dtype=np.uint8)
    parsing = (g.standard_normal((n_frames, 19, H, W)) * 4).astype(np.float32)
    sd = synth.synth_state_dict(load_keys("T"), 0)
    eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, "toonify", 256, dtype, dev)
    style = synth.synth_style(seed=5).to(dev)
    return frames, parsing, eng, style


def _expected(eng, style, frames, parsing, dev):
    """The reference loop's order of operations, one frame at a time, host-side pack / unpack."""
    out = []
    for f, p in zip(frames, parsing):
        x = torch.from_numpy(FO.pack_inputs(f[None], p[None])).to(dev)
        y = eng.forward(x, style, None, shared_style=True)
        out.append(FO.tensor2cv2(y[0].float().cpu().numpy()))
    return out


def test_video_driver_matches_frame_by_frame(dev):
    big = dev.type == "cuda"
    n, H, W = (11, 64, 96) if big else (4, 16, 24)
    frames, parsing, eng, style = _video_case(dev, n, H, W, 2, 2, torch.bfloat16)
    want = _expected(eng, style, frames, parsing, dev)
    for batch, depth in (((2, 2), (4, 1), (3, 3)) if big else ((3, 2),)):   # every in-flight slot builds its own plan
        got = {}
        vt = video.VideoToonifier(eng, style, None, batch_size=batch, bgr=True, depth=depth)
        order = []

        def sink(i, fr):
            order.append(i)
            got[i] = fr.copy()

        assert vt.run(((frames[i], parsing[i]) for i in range(n)), sink) == n
        assert order == list(range(n)), "frames must reach the sink in order"
        for i in range(n):
            assert got[i].shape == (4 * H, 4 * W, 3) and np.array_equal(got[i], want[i]), (batch, depth, i)
    # two ranks' shards, concatenated, are the video (frame-parallel; no communication)
    got = {}
    spans = [video.toonify_shard(eng, style, None, lambda i: (frames[i], parsing[i]), n,
                                 lambda i, fr: got.__setitem__(i, fr.copy()), batch_size=2, rank=r, world_size=2)
             for r in range(2)]
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == n
    assert all(np.array_equal(got[i], want[i]) for i in range(n))


def test_video_driver_computes_parsing_maps_on_the_gpu(dev):
    """Source yields no parsing maps + a parsing engine: x_p comes from BiSeNet on the device
    (style_transfer.py:170-172) and must equal parsing_maps() -> frame-by-frame forward."""
    from vtoonify_amd.bisenet import BiSeNetEngine
    big = dev.type == "cuda"
    n, H, W = (7, 64, 96) if big else (3, 32, 32)
    dtype = torch.bfloat16
    frames, _, eng, style = _video_case(dev, n, H, W, 2, 2, dtype)
    bsd = synth.synth_state_dict(load_keys("bisenet"), 0)
    par = BiSeNetEngine({k: v.to(dev) for k, v in bsd.items()}, 19, dtype, dev)
    want = []
    for f in frames:
        rgb = video.frame_pack(torch.from_numpy(f[None]).to(dev), None, bgr=True)
        xp = par.parsing_maps(rgb)
        x = video.frame_pack(torch.from_numpy(f[None]).to(dev), xp, bgr=True)
        want.append(video.frame_unpack(eng.forward(x, style, None, shared_style=True), bgr=True)[0].cpu().numpy())
    for batch, depth in ((2, 2), (3, 3)) if big else ((2, 2),):
        got = {}
        vt = video.VideoToonifier(eng, style, None, batch_size=batch, bgr=True, depth=depth, parsing_engine=par)
        assert vt.run(((frames[i], None) for i in range(n)), lambda i, fr: got.__setitem__(i, fr.copy())) == n
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (batch, depth, i)
