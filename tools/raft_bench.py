#!/usr/bin/env python
"""RAFT flow (vtoonify_amd.raft) + parsing-map fusion (vtoonify_amd.smooth) on one window of a video, the unit of work
of smooth_parsing_map.py:143-167: 2*window+1 frame pairs of 512x512 (the script's 2x-enlarged 256x256 crops), 20
refinement iterations, then warp + fusion + Downsample.

    python tools/raft_bench.py [--window 5] [--size 512] [--iters 20] [--reps 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vtoonify_amd import _lib, smooth, synth  # noqa: E402
from vtoonify_amd.raft import RAFT, raft_schema  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="arithmetic of the convolutions (correlation lookup, GRU state exchange and flow stay fp32)")
    ap.add_argument("--pairs", type=int, default=0, help="frame pairs per call (default: the whole window, 2*window+1)")
    a = ap.parse_args()
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    m = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False),
             compute_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    m.load_state_dict(synth.synth_state_dict({k: tuple(v) for k, v in raft_schema().items()}, 0))
    m = m.to(dev).eval()
    wn, S = 2 * a.window + 1, a.size
    g = torch.Generator().manual_seed(1)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 3, S + 2 * wn, S, generator=g), 5, stride=1, padding=2)
    Is = torch.stack([base[0, :, k:k + S] for k in range(wn)]).to(dev) * 2 - 1
    Ps = (torch.randn(wn, 19, S, S, generator=g) * 4).to(dev)
    wt = smooth.temporal_weights(a.window, dev)
    image1 = Is[a.window:a.window + 1].repeat(wn, 1, 1, 1)

    def flow():
        return m((image1 + 1) * 255.0 / 2, (Is + 1) * 255.0 / 2, iters=a.iters, test_mode=True)[1]

    def fuse(f):
        return smooth.fuse_window(Is[a.window].contiguous(), Is, Ps, f, wt, a.window)

    f = flow()
    fuse(f)
    torch.cuda.synchronize()
    if a.dtype != "fp32":   # deviation of the reduced-precision flow from the fp32 one on the same weights / frames
        m32 = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False))
        m32.load_state_dict(m.state_dict())
        m32 = m32.to(dev).eval()
        f32 = m32((image1 + 1) * 255.0 / 2, (Is + 1) * 255.0 / 2, iters=a.iters, test_mode=True)[1]
        print(f"flow_up {a.dtype} vs fp32: max |d| {float((f - f32).abs().max()):.4f} px, mean |d| "
              f"{float((f - f32).abs().mean()):.5f} px, max |flow| {float(f32.abs().max()):.2f} px")
        del m32
    tf, tu = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        f = flow()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        y = fuse(f)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tf.append(t1 - t0)
        tu.append(t2 - t1)
    tf.sort()
    tu.sort()
    print(f"RAFT {wn} pairs {S}x{S}, {a.iters} iterations: {1e3 * tf[len(tf) // 2]:.1f} ms "
          f"({1e3 * tf[len(tf) // 2] / wn:.2f} ms per pair, {a.dtype}, hipGraph replay {os.environ.get('VT_RAFT_GRAPH', '0') == '1'}); warp + fusion + Downsample of the window: "
          f"{1e3 * tu[len(tu) // 2]:.2f} ms; output {tuple(y.shape)}, finite {bool(torch.isfinite(y).all())}")


if __name__ == "__main__":
    main()
