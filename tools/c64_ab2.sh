#!/bin/bash
C=(python tools/conv_bench.py --only "=same 64 @512" --batch 4 --iters 30)
echo "plain:      $("${C[@]}" 2>/dev/null | grep '^same')"
echo "rgb:        $("${C[@]}" --rgb 2>/dev/null | grep '^same')"
for a in 34 35; do echo "rgb ABL $a: $(VT_RGB_ABLATE=$a "${C[@]}" --rgb 2>/dev/null | grep '^same')"; done
