#!/bin/bash
# conv_fullkw: cost of the AdaIN consumer / statistics producer / residual variants, and their ablations
C=(python tools/conv_bench.py --stream --only "=res 512->512 @32" --batch 4 --iters 50)
for a in 0 1 2 3 4 5 6 7; do echo "adain=$a: $("${C[@]}" --adain $a 2>/dev/null | grep '^res')"; done
for abl in 44 45; do for a in 1 3; do echo "ABL $abl adain=$a: $(VT_FULLKW_ABLATE=$abl "${C[@]}" --adain $a 2>/dev/null | grep '^res')"; done; done
for a in 2 3; do echo "ABL 46 adain=$a: $(VT_FULLKW_ABLATE=46 "${C[@]}" --adain $a 2>/dev/null | grep '^res')"; done
