"""PCIe-inclusive rate of the video driver (vtoonify_amd/video.py) and where the host time goes.
usage: python tools/video_bench.py [--batch 4] [--depth 2] [--frames 96]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import state_shapes  # noqa: E402
from vtoonify_amd import synth, video  # noqa: E402
from vtoonify_amd.engine import VToonifyEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=96)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(state_shapes("dualstylegan"), 0)
    eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, "dualstylegan", 256, torch.bfloat16, dev)
    style = synth.synth_style(seed=17).to(dev)
    H = W = a.size
    g = np.random.default_rng(0)
    frames = g.integers(0, 256, (8, H, W, 3), dtype=np.uint8)
    parsing = (g.standard_normal((8, 19, H, W)) * 4).astype(np.float32)
    # raw transfer rates of the staging buffers (pinned <-> device), HIP events on a side stream
    st = torch.cuda.Stream(dev)
    for name, shape, dt_ in (("parsing 4x19x256x256 f32", (4, 19, H, W), torch.float32),
                             ("frames out 4x1024x1024x3 u8", (4, 4 * H, 4 * W, 3), torch.uint8)):
        hbuf = torch.empty(shape, dtype=dt_, pin_memory=True)
        dbuf = torch.empty(shape, dtype=dt_, device=dev)
        for direction in ("h2d", "d2h"):
            with torch.cuda.stream(st):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(2):
                    (dbuf.copy_(hbuf, non_blocking=True) if direction == "h2d" else hbuf.copy_(dbuf, non_blocking=True))
                e0.record(st)
                for _ in range(10):
                    (dbuf.copy_(hbuf, non_blocking=True) if direction == "h2d" else hbuf.copy_(dbuf, non_blocking=True))
                e1.record(st)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"batch-sized {direction} {name}: {ms:.3f} ms = {hbuf.numel() * hbuf.element_size() / ms / 1e6:.1f} GB/s", flush=True)
    for batch, depth, with_p in ((1, 1, True), (1, 2, True), (1, 3, True), (1, 4, True), (2, 2, True), (2, 3, True), (4, 1, True),
                                 (4, 2, True), (4, 3, True), (8, 2, True), (8, 3, True)):
        vt = video.VideoToonifier(eng, style, 0.5, batch_size=batch, depth=depth)
        src = lambda n: ((frames[i % 8], parsing[i % 8] if with_p else None) for i in range(n))
        if not with_p:
            continue   # the engine needs 22 channels; kept for symmetry
        vt.run(src(2 * batch), lambda i, f: None)
        torch.cuda.synchronize()
        t = {"stage": 0.0, "submit": 0.0, "retire": 0.0}
        o_submit, o_retire = vt._submit, vt._retire

        def submit(slot, n):
            t0 = time.perf_counter(); o_submit(slot, n); t["submit"] += time.perf_counter() - t0

        def retire(slot, sink):
            t0 = time.perf_counter(); o_retire(slot, sink); t["retire"] += time.perf_counter() - t0

        vt._submit, vt._retire = submit, retire
        t0 = time.perf_counter()
        vt.run(src(a.frames), lambda i, f: None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # device-resident rate of the same batch for comparison
        x = video.frame_pack(torch.from_numpy(frames[:batch].repeat(1, 0)[:batch]).to(dev),
                             torch.from_numpy(parsing[:batch]).to(dev))
        for _ in range(3):
            eng.forward(x, style, 0.5, shared_style=True, use_graph=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            eng.forward(x, style, 0.5, shared_style=True, use_graph=True)
        torch.cuda.synchronize()
        res = 20 * batch / (time.perf_counter() - t1)
        print(f"batch {batch} depth {depth}: {a.frames / dt:7.1f} frames/s end to end ({res:7.1f} HBM-resident); "
              f"host s: total {dt:.3f} submit {t['submit']:.3f} retire(wait+sink) {t['retire']:.3f} "
              f"stage {dt - t['submit'] - t['retire']:.3f}", flush=True)


if __name__ == "__main__":
    main()
