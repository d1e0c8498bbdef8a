#!/bin/bash
# conv_upblur: tile width / double buffering at batch 4 on the two deepest levels
for only in "=up 512->512 @32->64" "=up 512->256 @64->128"; do
  for hint in 16 32; do for db in 99 4; do
    echo "hint $hint DB>=$db: $(VT_UPBLUR_DB=$db python tools/conv_bench.py --upblur --only "$only" --batch 4 --iters 50 --hint $hint 2>/dev/null | grep '^up')"
  done; done
done
