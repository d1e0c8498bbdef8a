#!/usr/bin/env python
"""Out-of-bounds WRITE detector for the engine's plan buffers: every buffer is allocated with a guard zone on either side filled
with a pattern; after a few frames the guards are checked (python tools/guard_bufs.py [T|D] B H W)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402

from conftest import load_keys  # noqa: E402
from vtoonify_amd import synth  # noqa: E402
from vtoonify_amd.engine import VToonifyEngine  # noqa: E402

bb, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
G = 8192   # guard bytes on either side
guards = []


def _buf(self, plan, name, shape, dtype=None):
    dtype = dtype or self.dtype
    esz = torch.empty((), dtype=dtype).element_size()
    n = 1
    for s in (shape if isinstance(shape, (tuple, list)) else (shape,)):
        n *= int(s)
    raw = torch.full((2 * G + n * esz + 64,), 0xA5, dtype=torch.uint8, device=self.device)
    off = G + (-(raw.data_ptr() + G)) % 64          # 64-byte aligned payload
    t = raw[off:off + n * esz].view(dtype).reshape(shape)
    plan.bufs[name] = t
    plan.bufs["__raw__" + name] = raw
    guards.append((name, raw, off, n * esz))
    return t


VToonifyEngine._buf = _buf
backbone = "toonify" if bb == "T" else "dualstylegan"
sd = synth.synth_state_dict(load_keys(bb), 0)
eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, backbone, 256, torch.bfloat16, dev)
style = synth.synth_style(seed=5).to(dev)
g = torch.Generator().manual_seed(1)
d_s = 0.6 if bb == "D" else None
for it in range(3):
    x = torch.randn(B, 22, H, W, generator=g).to(dev)
    eng.forward(x, style, d_s, shared_style=True, use_graph=False)
torch.cuda.synchronize()
bad = 0
for name, raw, off, nb in guards:
    lo, hi = raw[:off], raw[off + nb:]
    for side, z in (("below", lo), ("above", hi)):
        w = (z != 0xA5).nonzero()
        if w.numel():
            bad += 1
            idx = w.flatten()
            print(f"GUARD HIT {side} {name}: {idx.numel()} bytes, first at {int(idx[0]) - (off if side == 'below' else 0)}, last at "
                  f"{int(idx[-1]) - (off if side == 'below' else 0)} (payload {nb} bytes)")
print(bb, B, H, W, "buffers", len(guards), "guard hits", bad)
