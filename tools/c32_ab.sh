#!/bin/bash
# conv3x3_c32_kernel at 1024^2, batch 4: fused ToRGB ablations (VT_RGB_ABLATE 1 no rgb stores, 2 no skip loads, 3 no shuffles)
C=(python tools/conv_bench.py --only "=same 32 @1024" --batch 4 --iters 30)
echo "plain:   $("${C[@]}" 2>/dev/null | grep '^same')"
echo "rgb:     $("${C[@]}" --rgb 2>/dev/null | grep '^same')"
for a in 1 2 3; do echo "rgb ABL $a: $(VT_RGB_ABLATE=$a "${C[@]}" --rgb 2>/dev/null | grep '^same')"; done
