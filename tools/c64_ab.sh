#!/bin/bash
# conv3x3_c64_kernel (64 -> 64 @512x512, batch 4): phase ablations (VT_RGB_ABLATE: 31 no activation stores, 32 no tap loop,
# 33 no patch loads, 1 no rgb stores, 2 no skip loads, 3 no shuffles), with and without the fused ToRGB
C=(python tools/conv_bench.py --only "=same 64 @512" --batch 4 --iters 30)
echo "plain:      $("${C[@]}" 2>/dev/null | grep '^same')"
echo "rgb:        $("${C[@]}" --rgb 2>/dev/null | grep '^same')"
for a in 31 32 33 1 2 3; do echo "rgb ABL $a: $(VT_RGB_ABLATE=$a "${C[@]}" --rgb 2>/dev/null | grep '^same')"; done
for a in 31 32 33; do echo "plain ABL $a: $(VT_RGB_ABLATE=$a "${C[@]}" 2>/dev/null | grep '^same')"; done
echo "rows16 rgb: $(VT_C64_ROWS=16 "${C[@]}" --rgb 2>/dev/null | grep '^same')"
echo "pipe1 rgb:  $(VT_C64_PIPE=1 "${C[@]}" --rgb 2>/dev/null | grep '^same')"
echo "c32 rgb:    $(python tools/conv_bench.py --only "=same 32 @1024" --batch 4 --iters 30 --rgb 2>/dev/null | grep '^same')"
echo "c32 plain:  $(python tools/conv_bench.py --only "=same 32 @1024" --batch 4 --iters 30 2>/dev/null | grep '^same')"
