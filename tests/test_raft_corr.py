"""RAFT correlation lookup (vtoonify_amd/raft_corr.py, csrc/raft_corr.hip): the
reference's third native extension.  Golden = the reference's own pure-PyTorch CorrBlock
(tests/golden/make_golden_raft.py), which alt_cuda_corr is the memory-efficient form of.
Tolerance (fp32, relative to max|ref|): 2e-5 -- only the summation order over channels differs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from vtoonify_amd import raft_corr

TOL = 2e-5


def test_oracle_pinned_to_reference_corrblock():
    from oracle import raft_oracle as R
    d, _ = load_golden("raft_corr.npz")
    for name in ("a", "b"):
        L, r = (int(v) for v in d[name + "__cfg"])
        y = R.alternate_corr_block(d[name + "__f1"], d[name + "__f2"], d[name + "__coords"], L, r)
        assert y.shape == d[name + "__y"].shape
        assert rel_err(y, d[name + "__y"]) < TOL, name


def test_alternate_corr_block_matches_reference(dev):
    d, _ = load_golden("raft_corr.npz")
    for name in ("a", "b"):
        L, r = (int(v) for v in d[name + "__cfg"])
        f1, f2, c = (torch.from_numpy(d[name + k]).to(dev) for k in ("__f1", "__f2", "__coords"))
        y = raft_corr.AlternateCorrBlock(f1, f2, num_levels=L, radius=r)(c)
        assert tuple(y.shape) == d[name + "__y"].shape
        assert rel_err(y.cpu().numpy(), d[name + "__y"]) < TOL, name


def test_forward_has_the_extension_contract(dev):
    """alt_cuda_corr.forward(fmap1, fmap2, coords, r) -> [corr (B,1,(2r+1)^2,H,W)] (correlation.cpp:24-34),
    against the oracle's restatement of the CUDA kernel; errors like its CHECK_INPUT."""
    from oracle import raft_oracle as R
    from vtoonify_amd import _lib
    g = np.random.default_rng(2)
    f1 = g.standard_normal((2, 5, 7, 16)).astype(np.float32)
    f2 = g.standard_normal((2, 3, 4, 16)).astype(np.float32)     # a pyramid level: smaller than fmap1
    c = (g.standard_normal((2, 1, 5, 7, 2)) * 3 + 2).astype(np.float32)
    out = raft_corr.forward(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev), torch.from_numpy(c).to(dev), 2)
    assert isinstance(out, list) and len(out) == 1 and tuple(out[0].shape) == (2, 1, 25, 5, 7)
    assert rel_err(out[0].cpu().numpy(), R.corr_lookup(f1, f2, c, 2)) < TOL
    with pytest.raises(_lib.VtError):
        raft_corr.forward(torch.from_numpy(f1).to(dev).permute(0, 2, 1, 3), torch.from_numpy(f2).to(dev),
                          torch.from_numpy(c).to(dev), 2)       # not contiguous
    with pytest.raises(_lib.VtError):
        raft_corr.forward(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev), torch.from_numpy(c[:, :, :4]).to(dev), 2)
