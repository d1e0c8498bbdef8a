#!/usr/bin/env python
"""Determinism stress of the persistent / weights-resident patch kernels: the same convolution many times, on three streams at
once, against the one-workgroup-per-tile form (VT_PATCH_PIPE=1) computed once.  Prints the number of mismatching runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vtoonify_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
P = 100000000
cases = [(4, 512, 0, 64, 96, 512, P + 256128, False), (4, 128, 0, 256, 384, 128, P + 256128, True), (4, 256, 256, 128, 192, 256, P + 256128, False),
         (4, 64, 0, 256, 384, 64, P + 256064, True), (3, 128, 0, 100, 140, 64, P + 256064, False)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for N, c0, c1, H, W, cout, hint, rgb in cases:
    cin = c0 + c1
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(N, H, W, cin, generator=g).to(dev).to(dt)
    xa = x[..., :c0].contiguous()
    xb = x[..., c0:].contiguous() if c1 else None
    wp = (torch.randn(cout, 9, cin, generator=g) / (9 * cin) ** 0.5).to(dev).to(dt)
    b = torch.randn(cout, generator=g).to(dev)
    r = torch.randn(N, H, W, cout, generator=g).to(dev).to(dt)
    rgbw = (torch.randn(3, 1, cout, generator=g) / 8).to(dev).to(dt)
    skip = torch.randn(N, 3, H, W, generator=g).to(dev)
    rgbb = torch.randn(3, generator=g).to(dev)
    streams = [torch.cuda.Stream() for _ in range(3)]

    def run(out, rgb_out, stream=None):
        kw = dict(src1=xb, c1=c1, ld1=c1) if c1 else {}
        if rgb:
            kw.update(rgb_weight=rgbw, rgb_bias=rgbb, rgb_resid=skip, rgb_out=rgb_out)
        else:
            kw.update(resid=r, ld_res=cout, beta=0.25)
        K.conv2d(src0=xa, c0=c0, ld0=c0, n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3, pad=1, bias=b,
                 act=K.ACT_LRELU, gain=2 ** 0.5, out=out, ld_out=cout, dtype=K.dt_code(dt), tile_hint=hint, **kw)
    selfref = len(sys.argv) > 2 and sys.argv[2] == "self"   # reference = the same kernel, run once alone (determinism only)
    if not selfref:
        os.environ["VT_PATCH_PIPE"] = "1"
    ref, ref_rgb = torch.zeros(N, H, W, cout, dtype=dt, device=dev), torch.zeros(N, 3, H, W, device=dev)
    run(ref, ref_rgb)
    torch.cuda.synchronize()
    os.environ.pop("VT_PATCH_PIPE", None)
    outs = [(torch.zeros_like(ref), torch.zeros_like(ref_rgb)) for _ in streams]
    bad = 0
    for it in range(iters):
        for s, (o, ro) in zip(streams, outs):
            with torch.cuda.stream(s):
                o.fill_(float("nan")); ro.zero_()
                run(o, ro, s)
        torch.cuda.synchronize()
        for o, ro in outs:
            if not (torch.equal(o, ref) and torch.equal(ro, ref_rgb)):
                bad += 1
    print(f"N={N} cin={cin} {H}x{W} cout={cout} hint={hint % P} rgb={rgb}: {bad} mismatching runs of {3 * iters}")
