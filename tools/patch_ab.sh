#!/bin/bash
# patch-resident tile kernel: 256x128 tiles where the per-image heuristic takes 128x64, batch 4
for only in "=same 256 @128" "=fus2 512->256 @128" "=enc1.2 256->256 @128" "=enc2.2 512->512 @64" "=fus1 1024->512 @64"; do
  for hint in 0 101256128 101128128; do
    echo "hint $hint: $(python tools/conv_bench.py --stream --only "$only" --batch 4 --iters 30 --hint $hint 2>/dev/null | grep -v '^total\|^#' | tail -1)"
  done
done
