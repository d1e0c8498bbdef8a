#!/usr/bin/env python
"""Micro-benchmark of vt_conv2d on the conv shapes of one VToonify-D frame (22x256x256).

    python tools/conv_bench.py [--dtype bf16] [--hint SPLITK*1e6+BM*1e3+BN] [--only SUBSTR] [--iters 20]

Prints per-shape time, TFLOP/s (flops actually issued) and GB/s (operands once).  Used to
tune tiles / split-K and as the target of `rocprofv3 --pmc` runs.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vtoonify_amd import _lib, kernels as K  # noqa: E402

# (name, h, w, cin, cout, k, stride, dil, phases, out_layout)
SHAPES = [
    ("enc0.0 22->32 @256", 256, 256, 24, 32, 3, 1, 1, 1, "nhwc"),
    ("enc0.2 32->128 @256", 256, 256, 32, 128, 3, 1, 1, 1, "nhwc"),
    ("enc1.0 128->256 s2", 256, 256, 128, 256, 3, 2, 1, 1, "nhwc"),
    ("enc1.2 256->256 @128", 128, 128, 256, 256, 3, 1, 1, 1, "nhwc"),
    ("enc2.0 256->512 s2", 128, 128, 256, 512, 3, 2, 1, 1, "nhwc"),
    ("enc2.2 512->512 @64", 64, 64, 512, 512, 3, 1, 1, 1, "nhwc"),
    ("enc3.0 512->512 s2", 64, 64, 512, 512, 3, 2, 1, 1, "nhwc"),
    ("res 512->512 @32", 32, 32, 512, 512, 3, 1, 1, 1, "nhwc"),
    ("res 512->512 @8x32", 8, 32, 512, 512, 3, 1, 1, 1, "nhwc"),
    ("res 512->512 @16x32", 16, 32, 512, 512, 3, 1, 1, 1, "nhwc"),
    ("res 512->512 @32x64", 32, 64, 512, 512, 3, 1, 1, 1, "nhwc"),
    ("modres 512->512 @32 dil4", 32, 32, 512, 512, 3, 1, 4, 1, "nhwc"),
    ("fus0 1024->512 @32", 32, 32, 1024, 512, 3, 1, 1, 1, "nhwc"),
    ("mask0 1024->1 @32", 32, 32, 1024, 1, 3, 1, 1, 1, "nchw"),
    ("fskip0 520->3 @32", 32, 32, 520, 3, 3, 1, 1, 1, "nchw"),
    ("up 512->512 @32->64", 32, 32, 512, 512, 3, 1, 1, 4, "nhwc"),
    ("same 512 @64", 64, 64, 512, 512, 3, 1, 1, 1, "nhwc"),
    ("rgb 512->3 @64", 64, 64, 512, 3, 1, 1, 1, 1, "nchw"),
    ("fus1 1024->512 @64", 64, 64, 1024, 512, 3, 1, 1, 1, "nhwc"),
    ("mask1 1024->1 @64", 64, 64, 1024, 1, 3, 1, 1, 1, "nchw"),
    ("up 512->256 @64->128", 64, 64, 512, 256, 3, 1, 1, 4, "nhwc"),
    ("same 256 @128", 128, 128, 256, 256, 3, 1, 1, 1, "nhwc"),
    ("fus2 512->256 @128", 128, 128, 512, 256, 3, 1, 1, 1, "nhwc"),
    ("mask2 512->1 @128", 128, 128, 512, 1, 3, 1, 1, 1, "nchw"),
    ("up 256->128 @128->256", 128, 128, 256, 128, 3, 1, 1, 4, "nhwc"),
    ("same 128 @256", 256, 256, 128, 128, 3, 1, 1, 1, "nhwc"),
    ("fus3 256->128 @256", 256, 256, 256, 128, 3, 1, 1, 1, "nhwc"),
    ("mask3 256->1 @256", 256, 256, 256, 1, 3, 1, 1, 1, "nchw"),
    ("fskip3 136->3 @256", 256, 256, 136, 3, 3, 1, 1, 1, "nchw"),
    ("up 128->64 @256->512", 256, 256, 128, 64, 3, 1, 1, 4, "nhwc"),
    ("same 64 @512", 512, 512, 64, 64, 3, 1, 1, 1, "nhwc"),
    ("rgb 64->3 @512", 512, 512, 64, 3, 1, 1, 1, 1, "nchw"),
    ("up 64->32 @512->1024", 512, 512, 64, 32, 3, 1, 1, 4, "nhwc"),
    ("same 32 @1024", 1024, 1024, 32, 32, 3, 1, 1, 1, "nhwc"),
    ("rgb 32->3 @1024", 1024, 1024, 32, 3, 1, 1, 1, 1, "nchw"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--hint", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--nosplit", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the register-staged loader")
    ap.add_argument("--nopatch", action="store_true", help="disable the patch-resident kernel")
    ap.add_argument("--upblur", action="store_true",
                    help="run the up-sampling convs (phases 4 rows) as conv_transpose2d + LDS blur (vt_conv_desc.up_fir)")
    ap.add_argument("--stream", action="store_true",
                    help="attach the fragment-stream weights (whole-K kernel where eligible; add --hint 400000000 to force it)")
    ap.add_argument("--adain", type=int, default=0,
                    help="whole-K trunk convs: bit 0 = AdaIN consumer (in_tile_stats + in_gb), bit 1 = emit tile_stats, bit 2 = residual")
    ap.add_argument("--rgb", action="store_true", help="attach the fused ToRGB epilogue to the same-resolution convs")
    ap.add_argument("--lib", default="", help="another build of the library (same-box A/B of two .so files)")
    ap.add_argument("--sweep", default="", help="VAR=a,b,c: time every shape under each value of an environment switch the "
                    "library reads per call (same process, same buffers: a same-box A/B)")
    args = ap.parse_args()
    _lib.use_library(args.lib or _lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esz = 2 if dt == torch.bfloat16 else 4
    ws = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
    lib = _lib.lib()
    import ctypes as C
    tot = 0.0
    for name, h, w, cin, cout, k, stride, dil, phases, lay in SHAPES:
        if args.only and (args.only[1:] != name if args.only.startswith("=") else args.only not in name):
            continue
        n = args.batch
        pad = dil * (k // 2)
        ho, wo = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
        x = torch.randn(n, h, w, cin, device=dev).to(dt)
        upb = args.upblur and phases == 4
        if upb:
            phases = 1
        wt = (torch.randn(phases * cout, k * k, cin, device=dev) / (k * k * cin) ** 0.5).to(dt)
        bias = torch.randn(cout, device=dev)
        up2 = 2 if (phases == 4 or upb) else 1
        if lay == "nhwc":
            out = torch.empty(n, ho * up2, wo * up2, cout, device=dev, dtype=dt)
            kw = dict(out=out, ld_out=cout)
        else:
            out = torch.empty(n, cout, ho, wo, device=dev, dtype=torch.float32)
            kw = dict(out=out, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32)
        d = K.make_conv_desc(src0=x, c0=cin, ld0=cin, n=n, h=h, w=w, out_h=ho, out_w=wo, weight=wt, cout=cout,
                             kh=k, kw=k, stride=stride, pad=pad, dil=dil, phases=phases, bias=bias,
                             act=K.ACT_LRELU, gain=1.414, dtype=K.dt_code(dt), tile_hint=args.hint + (1000000000 if args.generic else 0) + (200000000 if args.nopatch else 0), **kw)
        if upb:
            k1 = torch.tensor([1.0, 3.0, 3.0, 1.0], device=dev)
            firt = (torch.outer(k1, k1) / 16.0).contiguous()
            d.up_fir = firt.data_ptr()
            d.out_h, d.out_w = 2 * ho, 2 * wo
        wst = None
        if args.stream and k == 3 and phases == 1 and not upb:
            wst = K.conv_weight_stream(wt)
            if wst is not None:
                d.weight_stream = wst.data_ptr()
        if args.rgb and name.startswith("same") and lay == "nhwc" and cout <= 128:
            rgbw = (torch.randn(3, 1, cout, device=dev) / cout ** 0.5).to(dt)
            rgbb = torch.randn(3, device=dev)
            rgbo = torch.randn(n, 3, ho, wo, device=dev)
            d.rgb_weight, d.rgb_bias = rgbw.data_ptr(), rgbb.data_ptr()
            d.rgb_resid = d.rgb_out = rgbo.data_ptr()
        keep = []
        if args.adain and name.startswith(("res", "modres")):
            nb = K.conv_tile_stats_bytes(n, h, w, dil, cout) // 4
            if args.adain & 1:   # records of a producer with dilation 1 over the same tensor
                prod = torch.zeros(K.conv_tile_stats_bytes(n, h, w, 1, cin) // 4, device=dev)
                dp = K.make_conv_desc(src0=x, c0=cin, ld0=cin, n=n, h=h, w=w, out_h=ho, out_w=wo, weight=wt, cout=cout, kh=k,
                                      kw=k, pad=1, bias=bias, act=K.ACT_LRELU, dtype=K.dt_code(dt), out=x.clone(), ld_out=cout,
                                      tile_stats=prod)
                dp.weight_stream = wst.data_ptr()
                _lib.check(lib.vt_conv2d(C.byref(dp), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "producer")
                gb = torch.randn(n, 2 * cin, device=dev)
                d.in_tile_stats, d.in_stats_dil, d.in_gb, d.in_ld_gb = prod.data_ptr(), 1, gb.data_ptr(), 2 * cin
                keep += [prod, gb]
            if args.adain & 2:
                ts = torch.zeros(nb, device=dev)
                d.tile_stats = ts.data_ptr()
                keep.append(ts)
            if args.adain & 4:
                rs = torch.randn(n, ho, wo, cout, device=dev).to(dt)
                d.resid, d.ld_res, d.beta = rs.data_ptr(), cout, 1.0
                keep.append(rs)
        if not args.nosplit:
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        tile = lib.vt_conv2d_tile(C.byref(d))
        if tile < 0:
            print(f"{name:<28} rejected: {lib.vt_last_error().decode()}")
            continue
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        var, vals = (args.sweep.split("=")[0], args.sweep.split("=")[1].split(",")) if args.sweep else ("", [""])
        for val in vals * (2 if args.sweep else 1):   # a sweep runs twice round: drift shows as a difference between rounds
            if var:
                os.environ[var] = val
            for _ in range(3):
                _lib.check(lib.vt_conv2d(C.byref(d), st), "conv")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                lib.vt_conv2d(C.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            m = n * ho * wo
            flops = 2.0 * m * (4 if upb else phases) * cout * k * k * cin   # MACs of the polyphase form, for comparison
            nbytes = x.numel() * esz + wt.numel() * esz + out.numel() * out.element_size()
            tot += us
            tag = f" {var}={val}" if var else ""
            print(f"{name:<28} tile {tile:>9d} M={m:8d} N={phases * cout:5d} K={k * k * cin:5d} {us:9.1f} us "
                  f"{flops / us / 1e6:8.1f} TF/s {nbytes / us / 1e3:8.1f} GB/s{tag}")
    print(f"total {tot:.1f} us")


if __name__ == "__main__":
    main()
