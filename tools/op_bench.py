#!/usr/bin/env python
"""Stand-alone benchmark of the two operators the reference implements as CUDA kernels --
upfirdn2d and fused_leaky_relu (model/stylegan/op) -- through the drop-in surface
(vtoonify_amd.op), at the tensor sizes they see in one VToonify-D frame at 22x256x256
(SURVEY.md 8a rows a13/a14, Appendix A).  Reports achieved ALGORITHMIC HBM bandwidth
(bytes = input + output, once each) against the MI355X HBM roofline (8 TB/s spec).

    python tools/op_bench.py [--dtype bf16|fp32|fp16] [--iters 20] [--json out.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vtoonify_amd import _lib, synth  # noqa: E402
from vtoonify_amd.op import fused_leaky_relu, upfirdn2d  # noqa: E402

PEAK = 8000.0  # GB/s


def timeit(fn, iters):
    """GPU time per call: `iters` calls captured in one hipGraph (the op surface allocates its output
    and goes through autograd.Function -- ~20 us of host work per call that a graph replay hides)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    dt = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[args.dtype]
    esz = torch.empty((), dtype=dt).element_size()
    k = synth.fir_kernel_2d().to(dev)
    rows = []
    # Blur after the transposed conv: (C, 2h+1, 2w+1) -> (C, 2h, 2w), k*4, pad (1,1)  (model.py:74-90)
    for c, s in ((512, 65), (256, 129), (128, 257), (64, 513), (32, 1025)):
        x = torch.randn(1, c, s, s, device=dev).to(dt)
        us = timeit(lambda: upfirdn2d(x, k * 4, pad=(1, 1)), args.iters)
        nb = (x.numel() + c * (s - 1) ** 2) * esz
        rows.append((f"upfirdn2d blur  ({c},{s},{s})", us, nb))
    # Upsample of the RGB skip: up=2, pad (2,1)  (model.py:32-50) -- fp32 planes in the reference
    for s in (32, 64, 128, 256, 512):
        x = torch.randn(1, 3, s, s, device=dev)
        us = timeit(lambda: upfirdn2d(x, k * 4, up=2, pad=(2, 1)), args.iters)
        rows.append((f"upfirdn2d up2   (3,{s},{s}) f32", us, (x.numel() + 3 * 4 * s * s) * 4))
    # Downsample (training / smooth_parsing_map): down=2, pad (1,1)
    x = torch.randn(1, 64, 512, 512, device=dev).to(dt)
    us = timeit(lambda: upfirdn2d(x, k, down=2, pad=(1, 1)), args.iters)
    rows.append(("upfirdn2d down2 (64,512,512)", us, (x.numel() + 64 * 256 * 256) * esz))
    # FusedLeakyReLU after every StyledConv / ConvLayer (op/fused_act.py:104-119)
    for c, s in ((512, 32), (512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)):
        x = torch.randn(1, c, s, s, device=dev).to(dt)
        b = torch.randn(c, device=dev).to(dt)
        us = timeit(lambda: fused_leaky_relu(x, b), args.iters)
        rows.append((f"fused_leaky_relu ({c},{s},{s})", us, (2 * x.numel() + c) * esz))
    out = []
    for name, us, nb in rows:
        gbs = nb / us / 1e3
        print(f"{name:<36} {us:9.1f} us {nb / 1e6:9.2f} MB {gbs:9.1f} GB/s  {100 * gbs / PEAK:5.1f}% of 8 TB/s")
        out.append({"op": name, "us": us, "bytes": nb, "gbs": gbs, "frac_of_hbm_peak": gbs / PEAK})
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"dtype": args.dtype, "peak_gbs": PEAK, "note": "GPU time per call, 20 calls per hipGraph replay",
                       "rows": out}, f, indent=1)


if __name__ == "__main__":
    main()
