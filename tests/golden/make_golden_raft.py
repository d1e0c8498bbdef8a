#!/usr/bin/env python3
"""Golden fixture for RAFT's correlation lookup from the REAL reference (authoring container only).

    python tests/golden/make_golden_raft.py     # writes tests/golden/raft_corr.npz

The reference's native form (alt_cuda_corr) is CUDA-only; its pure-PyTorch twin
model.raft.core.corr.CorrBlock (all-pairs volume + avg-pool pyramid + F.grid_sample, corr.py:12-60)
computes the same function and runs here.  Nothing is copied from the reference: only tensors it computes.
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from model.raft.core.corr import CorrBlock  # noqa: E402

torch.set_grad_enabled(False)


def main():
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, (B, C, H, W, L, r) in {"a": (2, 32, 16, 24, 4, 4), "b": (1, 64, 8, 8, 2, 3)}.items():
        f1 = torch.randn(B, C, H, W, generator=g)
        f2 = torch.randn(B, C, H, W, generator=g)
        ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
        coords = torch.stack([xs, ys], 0)[None].repeat(B, 1, 1, 1) + torch.randn(B, 2, H, W, generator=g) * 3.0
        coords[:, :, 0, 0] = torch.tensor([-7.3, 2.6])        # far outside
        coords[:, :, 1, 1] = torch.tensor([float(W) + 1.2, float(H) - 0.5])
        coords[:, :, 2, 2] = torch.tensor([3.0, 4.0])         # integer coordinates: zero fractional part
        y = CorrBlock(f1, f2, num_levels=L, radius=r)(coords)
        out.update({f"{name}__f1": f1.numpy(), f"{name}__f2": f2.numpy(), f"{name}__coords": coords.numpy(),
                    f"{name}__y": y.numpy(), f"{name}__cfg": np.array([L, r])})
        print(name, tuple(y.shape), float(y.abs().max()))
    path = os.path.join(HERE, "raft_corr.npz")
    np.savez_compressed(path, **out)
    print(f"wrote raft_corr.npz: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
