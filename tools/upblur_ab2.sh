#!/bin/bash
# conv_upblur: tall tiles (24 x 16 quads, 8 waves) vs 12-row tiles at batch 4
for only in "=up 512->512 @32->64" "=up 512->256 @64->128" "=up 256->128 @128->256" "=up 128->64 @256->512"; do
  for tall in 0 1; do
    echo "tall>=$tall: $(VT_UPBLUR_TALL=$tall VT_UPBLUR_DB=99 python tools/conv_bench.py --upblur --only "$only" --batch 4 --iters 50 --hint 32 2>/dev/null | grep '^up')"
  done
done
