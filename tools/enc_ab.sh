#!/bin/bash
# stride-2 encoder convs (1-D LDS-DMA kernel): tile shapes at batch 4
for only in "=enc1.0 128->256 s2" "=enc2.0 256->512 s2" "=enc3.0 512->512 s2"; do
  for hint in 0 64064 64128 128064 128128; do
    echo "hint $hint: $(python tools/conv_bench.py --only "$only" --batch 4 --iters 50 --hint $hint 2>/dev/null | grep '^enc')"
  done
done
