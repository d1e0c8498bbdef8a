#!/usr/bin/env python
"""Timings of the backward contractions of the drop-in operators (the reference's
op/conv2d_gradfix.py:134-223, op/upfirdn2d.py:12-87, op/fused_act.py:29-87) at generator-sized layers -- the passes
tests/test_grad_golden.py checks against reference-made gradients, here only timed:

    python tools/grad_bench.py [--reps 5] [--json OUT]

Per case: forward, backward (grad_input + grad_weight [+ grad_bias]) and, for the convs, the weight gradient alone, in
ms per call (HIP events around `reps` eager calls, best of 3) and the contraction's TFLOP/s (2 * MACs of the forward
for each of forward / grad_input / grad_weight).  fp32 tensors run the exact fp32 MFMA, bf16 tensors the bf16 MFMA."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vtoonify_amd import _lib, synth  # noqa: E402
from vtoonify_amd.op import conv2d_gradfix, fused_leaky_relu, upfirdn2d  # noqa: E402


EMU = "--emu" in sys.argv   # host-emulated kernels on tiny shapes: checks that every case of this script runs (CPU container)


def timeit(fn, reps):
    fn()
    if EMU:
        import time
        t0 = time.perf_counter()
        fn()
        return 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        best = ms if best is None else min(best, ms)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default="")
    ap.add_argument("--emu", action="store_true")
    a = ap.parse_args()
    if EMU:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from emu import build_emu
        _lib.use_library(build_emu.build())
        dev = torch.device("cpu")
    else:
        _lib.use_library(_lib.DEFAULT_LIB)
        dev = torch.device("cuda:0")
    sc = (lambda c, s: (min(c, 16), min(s, 9 if s % 2 else 8))) if EMU else (lambda c, s: (c, s))
    g = torch.Generator().manual_seed(3)
    rows = []

    def conv_case(name, n, cin, cout, h, transposed, dt):
        (cin, h), cout = sc(cin, h), sc(cout, h)[0]
        x = torch.randn(n, cin, h, h, generator=g).to(dev).to(dt).requires_grad_(True)
        wshape = (cin, cout, 3, 3) if transposed else (cout, cin, 3, 3)
        w = (torch.randn(*wshape, generator=g) / (3 * cin ** 0.5)).to(dev).to(dt).requires_grad_(True)
        if transposed:   # the up-sampling StyledConv's contraction (model.py:273-286)
            fwd = lambda: conv2d_gradfix.conv_transpose2d(x, w, stride=2, padding=0)  # noqa: E731
        else:
            fwd = lambda: conv2d_gradfix.conv2d(x, w, padding=1)  # noqa: E731
        y = fwd()
        go = torch.randn(y.shape, generator=g).to(dev).to(y.dtype)
        macs = n * cin * cout * 9 * (h * h)       # per contraction (transposed: input pixels x taps)

        def only(which):   # a Function's needs_input_grad follows requires_grad of its inputs: switch the other one off
            x.requires_grad_(which != "w")
            w.requires_grad_(which != "x")

        only("both")
        with torch.no_grad():
            t_f = timeit(lambda: fwd(), a.reps)
        only("x")
        t_in = timeit(lambda: torch.autograd.grad(fwd(), (x,), go), a.reps) - t_f
        only("w")
        t_w = timeit(lambda: torch.autograd.grad(fwd(), (w,), go), a.reps) - t_f
        only("both")
        tf = lambda ms: round(2 * macs / (ms * 1e-3) / 1e12, 1)  # noqa: E731
        rows.append({"op": name, "dtype": str(dt).split(".")[-1], "forward_ms": round(t_f, 3), "grad_input_ms": round(t_in, 3),
                     "grad_weight_ms": round(t_w, 3), "forward_tflops": tf(t_f), "grad_input_tflops": tf(t_in),
                     "grad_weight_tflops": tf(max(t_w, 1e-6))})

    k = synth.fir_kernel_2d().to(dev)
    for dt in (torch.float32, torch.bfloat16):
        conv_case("conv2d 3x3 (4,512,64,64)", 4, 512, 512, 64, False, dt)
        conv_case("conv2d 3x3 (4,128,256,256)", 4, 128, 128, 256, False, dt)
        conv_case("conv_transpose2d 3x3 s2 (4,512,32,32)", 4, 512, 512, 32, True, dt)
        conv_case("conv_transpose2d 3x3 s2 (4,128,128,128)", 4, 128, 128, 128, True, dt)
        for c, s in ((512, 65), (32, 1025)):
            c, s = sc(c, s)
            x = torch.randn(4, c, s, s, generator=g).to(dev).to(dt).requires_grad_(True)
            fwd = lambda: upfirdn2d(x, k * 4, pad=(1, 1))  # noqa: E731
            go = torch.randn(fwd().shape, generator=g).to(dev).to(dt)
            t_f = timeit(lambda: fwd(), a.reps)
            t_b = timeit(lambda: torch.autograd.grad(fwd(), (x,), go), a.reps) - t_f
            nb = 2 * x.numel() * x.element_size()
            rows.append({"op": f"upfirdn2d blur (4,{c},{s},{s})", "dtype": str(dt).split(".")[-1], "forward_ms": round(t_f, 3),
                         "grad_input_ms": round(t_b, 3), "forward_gbs": round(nb / t_f / 1e6, 1),
                         "grad_input_gbs": round(nb / max(t_b, 1e-6) / 1e6, 1)})
        for c, s in ((512, 64), (32, 1024)):
            c, s = sc(c, s)
            x = torch.randn(4, c, s, s, generator=g).to(dev).to(dt).requires_grad_(True)
            b = torch.randn(c, generator=g).to(dev).to(dt).requires_grad_(True)
            fwd = lambda: fused_leaky_relu(x, b)  # noqa: E731
            go = torch.randn(x.shape, generator=g).to(dev).to(dt)
            t_f = timeit(lambda: fwd(), a.reps)
            t_b = timeit(lambda: torch.autograd.grad(fwd(), (x, b), go), a.reps) - t_f
            nb = 2 * x.numel() * x.element_size()
            rows.append({"op": f"fused_leaky_relu (4,{c},{s},{s})", "dtype": str(dt).split(".")[-1], "forward_ms": round(t_f, 3),
                         "grad_input_ms": round(t_b, 3), "forward_gbs": round(nb / t_f / 1e6, 1),
                         "grad_input_gbs": round(nb / max(t_b, 1e-6) / 1e6, 1)})
    for r in rows:
        print(json.dumps(r))
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"protocol": f"{a.reps} eager calls between HIP events, best of 3; backward = (forward + backward) - forward",
                       "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
