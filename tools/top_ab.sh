#!/bin/bash
# top levels at batch 4: persistent conv_upblur (weights resident in LDS), uncapped registers, c64 vs the patch kernel
U=(python tools/conv_bench.py --upblur --batch 4 --iters 30 --hint 32)
for only in "=up 64->32 @512->1024" "=up 128->64 @256->512"; do
  echo "default:    $("${U[@]}" --only "$only" 2>/dev/null | grep '^up')"
  echo "persist:    $(VT_UPBLUR_PERSIST=1 "${U[@]}" --only "$only" 2>/dev/null | grep '^up')"
  echo "lb2=0:      $(VT_UPBLUR_LB2=0 "${U[@]}" --only "$only" 2>/dev/null | grep '^up')"
done
C=(python tools/conv_bench.py --only "=same 64 @512" --batch 4 --iters 30 --rgb)
echo "c64:        $("${C[@]}" 2>/dev/null | grep '^same')"
echo "patch:      $(VT_C64_KERNEL=0 "${C[@]}" 2>/dev/null | grep '^same')"
