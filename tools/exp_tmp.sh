bash tools/ab.sh env VT_C64_KERNEL 1 0 2
python - <<'PY'
import sys, time, torch, math
sys.path.insert(0, '.')
from vtoonify_amd import _lib, kernels as K
_lib.use_library(_lib.DEFAULT_LIB)
dev = torch.device('cuda:0')
# operator surface: conv_transpose2d(3x3, stride 2) 512 -> 512 at 32x32 (StyledConv(upsample) of the reference, one frame): parity tiles vs gather form
for N, cin, H, cout in ((1, 512, 32, 512), (4, 512, 32, 512), (1, 256, 128, 128)):
    x = torch.randn(N, H, H, cin, device=dev).bfloat16()
    w = (torch.randn(cout, 9, cin, device=dev) / math.sqrt(9 * cin)).bfloat16()
    z = torch.zeros(N, 2 * H + 1, 2 * H + 1, cout, device=dev, dtype=torch.bfloat16)
    for hint, name in ((0, "parity tiles"), (1000000000, "gather form")):
        kw = dict(src0=x, c0=cin, ld0=cin, n=N, h=H, w=H, out_h=2 * H + 1, out_w=2 * H + 1, weight=w, cout=cout, kh=3, kw=3, stride=2, pad=0,
                  transposed=1, out=z, ld_out=cout, dtype=K.VT_BF16, tile_hint=hint)
        for _ in range(3): K.conv2d(**kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): K.conv2d(**kw)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
        print(f"conv_transpose2d {cin}->{cout} @{H}x{H} batch {N}: {name:<13} {us:8.1f} us")
PY
