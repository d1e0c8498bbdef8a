"""BiSeNet face-parsing network on the MI355X kernels -- SURVEY.md section 8f rank 2.

Reference: model/bisenet/model.py:13-254 (ConvBNReLU, BiSeNetOutput, AttentionRefinementModule,
ContextPath, FeatureFusionModule, BiSeNet) and model/bisenet/resnet.py:14-80 (BasicBlock,
Resnet18).  The video loop calls it on every batch at TWICE the frame resolution
(style_transfer.py:66-68, 171-172): 19 of the 22 input channels of VToonify.forward come from it.

`BiSeNet` below keeps the reference's constructor and state_dict schema (191 entries for
n_classes=19), so `parsingpredictor.load_state_dict(torch.load(faceparsing_path))` works
unchanged; the arithmetic runs in `BiSeNetEngine`, a static launch plan over libvtoonify_amd.so:
  * every Conv2d is one vt_conv2d launch with the FOLLOWING eval-mode BatchNorm folded into its
    weights/bias at load time and ReLU fused in the epilogue; the ReLU that follows the shortcut
    add of a BasicBlock is the conv's `post_relu`; torch.cat of the FeatureFusionModule never
    materialises (two-source loader);
  * global average pools are vt_channel_mean; the 1x1 convs on pooled vectors (conv_avg,
    conv_atten + bn_atten, ffm.conv1/conv2) are vt_linear launches (ReLU / sigmoid activations);
  * ARM product, "+ avg_up" / "+ feat32_up" and the nearest up-sampling that follows are ONE
    vt_gate_add_nearest pass; FFM's feat * atten + feat is the same kernel with gate = atten + 1;
  * the three class maps are planar fp32 (19 channels) and are resized by vt_resize_bilinear.
`parsing_maps` is the reference's pre/post-processing around the net fused with it: bilinear x2 of
the frame (align_corners=False) * 2 -> net -> only the pixels that the nearest x0.5 keeps are
evaluated by the final align_corners=True resize.
There is no eager-PyTorch fallback; weight folding at load time is parameter preprocessing.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from . import kernels as K
from ._lib import ACT_LRELU, ACT_NONE, ACT_SIGMOID, OUT_NCHW, OUT_NHWC

BN_EPS = 1e-5


# ------------------------------------------------------------------------- parameter schema
class _ConvBNReLU(nn.Module):
    def __init__(self, cin, cout, ks=3, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, ks, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(cout)


class _Output(nn.Module):
    def __init__(self, cin, mid, n_classes):
        super().__init__()
        self.conv = _ConvBNReLU(cin, mid)
        self.conv_out = nn.Conv2d(mid, n_classes, 1, bias=False)


class _ARM(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _ConvBNReLU(cin, cout)
        self.conv_atten = nn.Conv2d(cout, cout, 1, bias=False)
        self.bn_atten = nn.BatchNorm2d(cout)


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        if cin != cout or stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


class _Resnet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        for i, (cin, cout, stride) in enumerate(((64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)), 1):
            setattr(self, f"layer{i}", nn.Sequential(_BasicBlock(cin, cout, stride), _BasicBlock(cout, cout, 1)))


class _ContextPath(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnet = _Resnet18()
        self.arm16 = _ARM(256, 128)
        self.arm32 = _ARM(512, 128)
        self.conv_head32 = _ConvBNReLU(128, 128)
        self.conv_head16 = _ConvBNReLU(128, 128)
        self.conv_avg = _ConvBNReLU(512, 128, 1, 1, 0)


class _FFM(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.convblk = _ConvBNReLU(cin, cout, 1, 1, 0)
        self.conv1 = nn.Conv2d(cout, cout // 4, 1, bias=False)
        self.conv2 = nn.Conv2d(cout // 4, cout, 1, bias=False)


class BiSeNet(nn.Module):
    """Same constructor / state_dict / forward contract as model.bisenet.model.BiSeNet (eval mode)."""

    def __init__(self, n_classes=19, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.n_classes = n_classes
        self.compute_dtype = compute_dtype
        self.cp = _ContextPath()
        self.ffm = _FFM(256, 256)
        self.conv_out = _Output(256, 256, n_classes)
        self.conv_out16 = _Output(128, 64, n_classes)
        self.conv_out32 = _Output(128, 64, n_classes)
        self._engine: Optional[BiSeNetEngine] = None
        self.requires_grad_(False)
        self.eval()

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def engine(self) -> "BiSeNetEngine":
        if self._engine is None:
            dev = next(self.parameters()).device
            self._engine = BiSeNetEngine(self.state_dict(), self.n_classes, self.compute_dtype, dev)
        return self._engine

    def forward(self, x):
        """(B,3,H,W) -> (feat_out, feat_out16, feat_out32), each (B,n_classes,H,W) fp32 (model.py:241-254)."""
        return self.engine().forward(x)

    def parsing_maps(self, frames):
        """style_transfer.py:171-172 for frames (B,3,H,W) in [-1,1]: (B,n_classes,H,W) fp32."""
        return self.engine().parsing_maps(frames)


# ------------------------------------------------------------------------------- the engine
class BiSeNetEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], n_classes: int = 19,
                 dtype: torch.dtype = torch.bfloat16, device: Optional[torch.device] = None):
        assert dtype in (torch.bfloat16, torch.float32)
        self.dtype, self.dt = dtype, K.dt_code(dtype)
        self.device = device or next(iter(state_dict.values())).device
        if self.device.type != "cuda" and not _lib.is_emulation():
            raise _lib.VtError("BiSeNetEngine needs a GPU device (no CPU path)")
        self.lib = _lib.lib()
        self.n_classes = n_classes
        self.sd = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in state_dict.items()
                   if v.dtype.is_floating_point}
        self._plans: Dict[tuple, dict] = {}
        self._pack()

    # -- load-time parameter preprocessing ------------------------------------------------
    def _bn_affine(self, prefix):
        sd = self.sd
        scale = sd[prefix + "weight"] / torch.sqrt(sd[prefix + "running_var"] + BN_EPS)
        return scale.contiguous(), (sd[prefix + "bias"] - sd[prefix + "running_mean"] * scale).contiguous()

    def _pack(self):
        sd, T = self.sd, self.dtype
        self.w: Dict[str, torch.Tensor] = {}
        self.b: Dict[str, torch.Tensor] = {}

        def conv_bn(name, wkey, bnprefix, as_matrix=False):
            sc, sh = self._bn_affine(bnprefix)
            w = (sd[wkey] * sc.reshape(-1, 1, 1, 1)).contiguous()
            self.w[name] = w.reshape(w.shape[0], -1).contiguous() if as_matrix else K.pack_conv_weight(w, out_dtype=T)
            self.b[name] = sh

        r = "cp.resnet."
        conv_bn("stem", r + "conv1.weight", r + "bn1.")
        for li in range(1, 5):
            for bi in range(2):
                p = f"{r}layer{li}.{bi}."
                conv_bn(p + "c1", p + "conv1.weight", p + "bn1.")
                conv_bn(p + "c2", p + "conv2.weight", p + "bn2.")
                if p + "downsample.0.weight" in sd:
                    conv_bn(p + "ds", p + "downsample.0.weight", p + "downsample.1.")
        for a in ("arm16", "arm32"):
            conv_bn(f"cp.{a}.conv", f"cp.{a}.conv.conv.weight", f"cp.{a}.conv.bn.")
            conv_bn(f"cp.{a}.atten", f"cp.{a}.conv_atten.weight", f"cp.{a}.bn_atten.", as_matrix=True)
        conv_bn("cp.head32", "cp.conv_head32.conv.weight", "cp.conv_head32.bn.")
        conv_bn("cp.head16", "cp.conv_head16.conv.weight", "cp.conv_head16.bn.")
        conv_bn("cp.avg", "cp.conv_avg.conv.weight", "cp.conv_avg.bn.", as_matrix=True)
        conv_bn("ffm.blk", "ffm.convblk.conv.weight", "ffm.convblk.bn.")
        self.w["ffm.fc1"] = sd["ffm.conv1.weight"].reshape(sd["ffm.conv1.weight"].shape[0], -1).contiguous()
        self.w["ffm.fc2"] = sd["ffm.conv2.weight"].reshape(sd["ffm.conv2.weight"].shape[0], -1).contiguous()
        for o in ("conv_out", "conv_out16", "conv_out32"):
            conv_bn(o + ".conv", o + ".conv.conv.weight", o + ".conv.bn.")
            self.w[o + ".out"] = K.pack_conv_weight(sd[o + ".conv_out.weight"], out_dtype=T)

    # -- plan ------------------------------------------------------------------------------
    def _build(self, B, H, W, mode):
        """mode 'net': input = network input (B,3,H,W), outputs = 3 class maps at (H,W).
        mode 'maps': input = frames (B,3,H,W) in [-1,1]; the net runs at (2H,2W); output = head 0
        sampled at every second pixel = (B,n_classes,H,W)."""
        lib, dt, T = self.lib, self.dt, self.dtype
        dev, f32 = self.device, torch.float32
        plan = {"ops": [], "keep": [], "bufs": {}, "convs": [], "graph": None}
        ops, keep, bufs = plan["ops"], plan["keep"], plan["bufs"]
        nc = self.n_classes

        def buf(name, shape, dtype=None, zero=False):
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype or T, device=dev)
            bufs[name] = t
            return t

        def conv(**kw):
            d = K.make_conv_desc(dtype=dt, **kw)
            keep.append(d)
            plan["convs"].append(d)
            ops.append((lib.vt_conv2d, (C.byref(d),)))

        def relu_conv(name, src, cin, h, w, cout, out, k=3, stride=1, pad=1, **kw):
            ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
            conv(src0=src, c0=cin, ld0=cin, n=B, h=h, w=w, out_h=ho, out_w=wo, weight=self.w[name], cout=cout,
                 kh=k, kw=k, stride=stride, pad=pad, bias=self.b[name], act=ACT_LRELU, slope=0.0, out=out,
                 ld_out=cout, **kw)
            return ho, wo

        def linear(y, x, Wm, b, in_dim, out_dim, act):
            ops.append((lib.vt_linear, (C.c_void_p(y.data_ptr()), out_dim, C.c_void_p(x.data_ptr()), in_dim,
                                        C.c_void_p(Wm.data_ptr()), C.c_void_p(b.data_ptr() if b is not None else 0),
                                        B, in_dim, out_dim, 1.0, 1.0, act, 0.0, 1.0)))

        ws_bytes = 16

        def mean(dst, x, hw, c):
            nonlocal ws_bytes
            ws_bytes = max(ws_bytes, K.instnorm_ws_bytes(B, hw, c))
            ops.append(("mean", (dst, x, c, B, hw, c)))

        x_in = buf("x_in", (B, 3, H, W), f32)
        if mode == "maps":
            Hn, Wn = 2 * H, 2 * W
            x0 = buf("x0", (B, Hn, Wn, 8), zero=True)   # channels 3..7 stay zero
            # 2 * F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
            ops.append((lib.vt_resize_bilinear, (C.c_void_p(x0.data_ptr()), OUT_NHWC, 8, dt, C.c_void_p(x_in.data_ptr()),
                                                 B, 3, H, W, Hn, Wn, 0, 1, Hn, Wn, 2.0)))
        else:
            Hn, Wn = H, W
            x0 = buf("x0", (B, Hn, Wn, 8))
            ops.append((lib.vt_nchw_to_nhwc, (C.c_void_p(x0.data_ptr()), 8, C.c_void_p(x_in.data_ptr()), B, 3,
                                              Hn * Wn, K.VT_F32, dt)))
        # ---- Resnet18 (resnet.py:68-77) ---------------------------------------------------
        h, w = (Hn + 6 - 7) // 2 + 1, (Wn + 6 - 7) // 2 + 1
        stem = buf("stem", (B, h, w, 64))
        relu_conv("stem", x0, 8, Hn, Wn, 64, stem, k=7, stride=2, pad=3)
        hp, wp = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        cur = buf("pool", (B, hp, wp, 64))
        ops.append((lib.vt_maxpool2d, (C.c_void_p(cur.data_ptr()), C.c_void_p(stem.data_ptr()), B, h, w, 64, 3, 2, 1, dt)))
        h, w, cc = hp, wp, 64
        feats = {}
        for li, (cout, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), 1):
            for bi in range(2):
                p = f"cp.resnet.layer{li}.{bi}."
                st = stride if bi == 0 else 1
                ho, wo = (h + 2 - 3) // st + 1, (w + 2 - 3) // st + 1
                t1 = buf(p + "t1", (B, ho, wo, cout))
                relu_conv(p + "c1", cur, cc, h, w, cout, t1, stride=st)
                if p + "ds" in self.w:
                    sc = buf(p + "sc", (B, ho, wo, cout))
                    conv(src0=cur, c0=cc, ld0=cc, n=B, h=h, w=w, out_h=ho, out_w=wo, weight=self.w[p + "ds"], cout=cout,
                         kh=1, kw=1, stride=st, pad=0, bias=self.b[p + "ds"], out=sc, ld_out=cout)
                else:
                    sc = cur
                out = buf(p + "out", (B, ho, wo, cout))
                # relu(shortcut + bn2(conv2(.)))  (resnet.py:44-47)
                conv(src0=t1, c0=cout, ld0=cout, n=B, h=ho, w=wo, out_h=ho, out_w=wo, weight=self.w[p + "c2"],
                     cout=cout, kh=3, kw=3, pad=1, bias=self.b[p + "c2"], beta=1.0, resid=sc, ld_res=cout,
                     post_relu=1, out=out, ld_out=cout)
                cur, cc, h, w = out, cout, ho, wo
            feats[li] = (cur, cc, h, w)
        (f8, c8, h8, w8), (f16, c16, h16, w16), (f32_, c32, h32, w32) = feats[2], feats[3], feats[4]
        # ---- ContextPath (model.py:108-121) -----------------------------------------------
        m32 = buf("m32", (B, c32), f32)
        mean(m32, f32_, h32 * w32, c32)
        avg = buf("avg", (B, 128), f32)
        linear(avg, m32, self.w["cp.avg"], self.b["cp.avg"], c32, 128, ACT_LRELU)      # conv_avg: 1x1 + BN + ReLU

        def arm(tag, src, cin, hh, ww, add_vec, add, out_h, out_w):
            feat = buf(f"{tag}.feat", (B, hh, ww, 128))
            relu_conv(f"cp.{tag}.conv", src, cin, hh, ww, 128, feat)
            mu = buf(f"{tag}.mean", (B, 128), f32)
            mean(mu, feat, hh * ww, 128)
            gate = buf(f"{tag}.gate", (B, 128), f32)
            linear(gate, mu, self.w[f"cp.{tag}.atten"], self.b[f"cp.{tag}.atten"], 128, 128, ACT_SIGMOID)
            up = buf(f"{tag}.up", (B, out_h, out_w, 128))
            ops.append((lib.vt_gate_add_nearest, (C.c_void_p(up.data_ptr()), C.c_void_p(feat.data_ptr()),
                                                  C.c_void_p(gate.data_ptr()),
                                                  C.c_void_p(add_vec.data_ptr() if add_vec is not None else 0),
                                                  C.c_void_p(add.data_ptr() if add is not None else 0),
                                                  B, hh, ww, 128, out_h, out_w, dt)))
            return up

        up32 = arm("arm32", f32_, c32, h32, w32, avg, None, h16, w16)       # (arm32 + avg_up) -> nearest to 1/16
        cp16 = buf("cp16", (B, h16, w16, 128))
        relu_conv("cp.head32", up32, 128, h16, w16, 128, cp16)              # feat32_up
        up16 = arm("arm16", f16, c16, h16, w16, None, cp16, h8, w8)         # (arm16 + feat32_up) -> nearest to 1/8
        cp8 = buf("cp8", (B, h8, w8, 128))
        relu_conv("cp.head16", up16, 128, h8, w8, 128, cp8)                 # feat16_up
        # ---- FeatureFusionModule (model.py:197-208): cat[feat_res8, feat_cp8] -----------------
        fb = buf("ffm.feat", (B, h8, w8, 256))
        conv(src0=f8, c0=c8, ld0=c8, src1=cp8, c1=128, ld1=128, n=B, h=h8, w=w8, out_h=h8, out_w=w8,
             weight=self.w["ffm.blk"], cout=256, kh=1, kw=1, pad=0, bias=self.b["ffm.blk"], act=ACT_LRELU, slope=0.0,
             out=fb, ld_out=256)
        fm = buf("ffm.mean", (B, 256), f32)
        mean(fm, fb, h8 * w8, 256)
        f1 = buf("ffm.f1", (B, 64), f32)
        linear(f1, fm, self.w["ffm.fc1"], None, 256, 64, ACT_LRELU)
        fg = buf("ffm.gate", (B, 256), f32)
        linear(fg, f1, self.w["ffm.fc2"], None, 64, 256, ACT_SIGMOID)
        fuse = buf("ffm.out", (B, h8, w8, 256))
        ops.append((lib.vt_gate_add_nearest, (C.c_void_p(fuse.data_ptr()), C.c_void_p(fb.data_ptr()),
                                              C.c_void_p(fg.data_ptr()), C.c_void_p(0), C.c_void_p(fb.data_ptr()),
                                              B, h8, w8, 256, h8, w8, dt)))              # feat * atten + feat
        # ---- output heads (model.py:43-46, 248-254) -----------------------------------------
        heads = [("conv_out", fuse, 256, 256, h8, w8)]
        if mode == "net":
            heads += [("conv_out16", cp8, 128, 64, h8, w8), ("conv_out32", cp16, 128, 64, h16, w16)]
        outs = []
        for name, src, cin, mid, hh, ww in heads:
            t = buf(name + ".mid", (B, hh, ww, mid))
            relu_conv(name + ".conv", src, cin, hh, ww, mid, t)
            logits = buf(name + ".logits", (B, nc, hh, ww), f32)
            conv(src0=t, c0=mid, ld0=mid, n=B, h=hh, w=ww, out_h=hh, out_w=ww, weight=self.w[name + ".out"], cout=nc,
                 kh=1, kw=1, pad=0, out=logits, ld_out=0, out_layout=OUT_NCHW, out_dtype=K.VT_F32)
            if mode == "maps":   # F.interpolate(., (2H,2W), bilinear, align_corners=True)[:, :, ::2, ::2]
                y = buf(name + ".y", (B, nc, H, W), f32)
                ops.append((lib.vt_resize_bilinear, (C.c_void_p(y.data_ptr()), OUT_NCHW, 0, K.VT_F32,
                                                     C.c_void_p(logits.data_ptr()), B, nc, hh, ww, Hn, Wn, 1, 2, H, W,
                                                     1.0)))
            else:
                y = buf(name + ".y", (B, nc, Hn, Wn), f32)
                ops.append((lib.vt_resize_bilinear, (C.c_void_p(y.data_ptr()), OUT_NCHW, 0, K.VT_F32,
                                                     C.c_void_p(logits.data_ptr()), B, nc, hh, ww, Hn, Wn, 1, 1, Hn, Wn,
                                                     1.0)))
            outs.append(y)
        plan["outs"] = outs
        plan["taps"] = {"res8": (f8, c8, h8, w8), "cp8": (cp8, 128, h8, w8), "cp16": (cp16, 128, h16, w16)}
        bufs["partials"] = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        need = 0
        for d in plan["convs"]:
            b = int(lib.vt_conv2d_ws_bytes(C.byref(d)))
            if b < 0:
                raise _lib.VtError(f"vt_conv2d descriptor rejected: {lib.vt_last_error().decode()}")
            need = max(need, b)
        if need:
            ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
            bufs["splitk_ws"] = ws
            for d in plan["convs"]:
                d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), need
        return plan

    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def _launch(self, plan):
        st = self._stream()
        part = plan["bufs"]["partials"]
        for fn, args in plan["ops"]:
            if fn == "mean":
                mean, x, ld, n, hw, c = args
                rc = self.lib.vt_channel_mean(C.c_void_p(mean.data_ptr()), C.c_void_p(x.data_ptr()), ld, n, hw, c,
                                              C.c_void_p(part.data_ptr()), self.dt, st)
            else:
                rc = fn(*args, st)
            if rc != 0:
                raise _lib.VtError(f"BiSeNet plan op failed (code {rc}): {self.lib.vt_last_error().decode()}")

    def _run(self, x, mode, use_graph, lane):
        if x.ndim != 4 or x.shape[1] != 3:
            raise _lib.VtError("BiSeNet input must be (B,3,H,W)")
        if x.device != self.device and not (x.device.type == self.device.type == "cpu"):
            raise _lib.VtError(f"input on {x.device}, engine on {self.device}")
        B, _, H, W = x.shape
        key = (B, H, W, mode, lane)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._build(B, H, W, mode)
            self._plans[key] = plan
        plan["bufs"]["x_in"].copy_(x.detach())
        if use_graph and self.device.type == "cuda":
            if plan["graph"] is None:
                self._launch(plan)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch(plan)
                plan["graph"] = g
            plan["graph"].replay()
        else:
            self._launch(plan)
        return plan

    @torch.no_grad()
    def forward(self, x: torch.Tensor, use_graph: bool = False, taps: bool = False, lane: int = 0):
        plan = self._run(x, "net", use_graph, lane)
        outs = tuple(o.clone() for o in plan["outs"])
        if taps:
            B = x.shape[0]
            res = {k: K.nhwc_to_nchw(t, c, B, c, h, w, self.dtype, torch.float32, self.device, t)
                   for k, (t, c, h, w) in plan["taps"].items()}
            return outs, res
        return outs

    @torch.no_grad()
    def parsing_maps(self, frames: torch.Tensor, use_graph: bool = False, lane: int = 0, out=None):
        """x_p of style_transfer.py:171-172 (before the /16): (B,3,H,W) in [-1,1] -> (B,19,H,W) fp32."""
        plan = self._run(frames, "maps", use_graph, lane)
        y = plan["outs"][0]
        if out is not None:
            out.copy_(y)
            return out
        return y.clone()

    __call__ = forward
