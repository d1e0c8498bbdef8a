"""Flow warp + temporal fusion of parsing maps (vtoonify_amd/smooth.py, csrc/flow_ops.hip)
against tensors produced by the reference's OWN source lines (smooth_parsing_map.py:37-75,155-166 executed by
tests/golden/make_golden_smooth.py).  Tolerance, fp32 relative to max|ref|: 2e-5 (the fusion sums 2w+1 products in
a different association: sum(P w) / sum(w) instead of sum(P (w / sum(w)))); masks bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from vtoonify_amd import smooth

TOL = 2e-5


def _clips(d):
    for name in ("a", "b"):
        yield name, {k: d[f"{name}__{k}"] for k in ("Is", "Ps", "flows", "wt", "fused", "parse", "cfg")}


def test_oracle_pinned_to_reference_lines():
    from oracle import smooth_oracle as S
    d, _ = load_golden("smooth.npz")
    o, m = S.warp(d["w__x"], d["w__flo"])
    assert np.array_equal(np.broadcast_to(m, d["w__mask"].shape), d["w__mask"])
    assert rel_err(o, d["w__out"]) < 1e-6
    for name, c in _clips(d):
        w = int(c["cfg"][0])
        assert rel_err(S.temporal_weights(w), c["wt"]) < 1e-6      # (numpy's exp and torch's differ in the last bit)
        Is_ = np.concatenate([c["Is"][:w], c["Is"], c["Is"][-w:]], 0)
        Ps_ = np.concatenate([c["Ps"][:w], c["Ps"], c["Ps"][-w:]], 0)
        for ii in range(c["Is"].shape[0]):
            i = ii + w
            f = S.fuse_window(Is_[i], Is_[i - w:i + w + 1], Ps_[i - w:i + w + 1], c["flows"][ii], c["wt"], w)
            assert rel_err(f, c["fused"][ii]) < TOL, (name, ii)
            assert rel_err(S.downsample(f[None])[0], c["parse"][ii]) < TOL, (name, ii)


def test_warp_matches_reference(dev):
    d, _ = load_golden("smooth.npz")
    x, flo = torch.from_numpy(d["w__x"]).to(dev), torch.from_numpy(d["w__flo"]).to(dev)
    o, m = smooth.warp(x, flo)
    assert tuple(m.shape) == d["w__mask"].shape and np.array_equal(m.cpu().numpy(), d["w__mask"])
    assert rel_err(o.cpu().numpy(), d["w__out"]) < 1e-6
    from vtoonify_amd import _lib
    with pytest.raises(_lib.VtError):
        smooth.warp(x, flo[:, :1])


def test_fusion_matches_reference(dev):
    d, _ = load_golden("smooth.npz")
    for name, c in _clips(d):
        w = int(c["cfg"][0])
        Is, Ps = torch.from_numpy(c["Is"]).to(dev), torch.from_numpy(c["Ps"]).to(dev)
        flows = torch.from_numpy(c["flows"]).to(dev)
        assert np.array_equal(smooth.temporal_weights(w).numpy(), c["wt"])
        calls = []

        def flow_fn(image1, image2):      # stands for raft_model(image1, image2, iters=20, test_mode=True)[1]
            assert tuple(image1.shape) == tuple(image2.shape) == (2 * w + 1, 3) + tuple(Is.shape[2:])
            calls.append(1)
            return flows[len(calls) - 1]

        y = smooth.smooth_parsing_maps(Is, Ps, flow_fn, w)
        assert tuple(y.shape) == c["parse"].shape and len(calls) == Is.shape[0]
        assert rel_err(y.cpu().numpy(), c["parse"]) < TOL, name
        # the un-decimated fusion of one frame
        Is_ = torch.cat((Is[:w], Is, Is[-w:]), 0)
        Ps_ = torch.cat((Ps[:w], Ps, Ps[-w:]), 0)
        i = 1 + w
        f = smooth.fuse_window(Is_[i].contiguous(), Is_[i - w:i + w + 1].contiguous(), Ps_[i - w:i + w + 1].contiguous(),
                               flows[1].contiguous(), smooth.temporal_weights(w, dev), down_kernel=False)
        assert rel_err(f[0].cpu().numpy(), c["fused"][1]) < TOL, name


@pytest.mark.gpu
def test_fusion_at_video_size_vs_oracle():
    """512x512 frames (the reference enlarges 256x256 crops 2x, smooth_parsing_map.py:127), window 2, vs the oracle."""
    from oracle import smooth_oracle as S
    from vtoonify_amd import _lib
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    g = np.random.default_rng(4)
    w, H, W = 2, 512, 512
    image2 = np.tanh(g.standard_normal((2 * w + 1, 3, H, W))).astype(np.float32)
    P = (g.standard_normal((2 * w + 1, 19, H, W)) * 4).astype(np.float32)
    flow = (g.standard_normal((2 * w + 1, 2, H, W)) * 3).astype(np.float32)
    wt = S.temporal_weights(w)
    want = S.downsample(S.fuse_window(image2[w], image2, P, flow, wt, w)[None])
    got = smooth.fuse_window(*(torch.from_numpy(a).to(dev) for a in (image2[w], image2, P, flow, wt)))
    assert rel_err(got.cpu().numpy(), want) < TOL
