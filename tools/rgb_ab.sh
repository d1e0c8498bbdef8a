#!/bin/bash
# fused ToRGB epilogue vs plain conv (+ what a separate ToRGB launch costs), batch 4
for only in "=same 128 @256" "=same 64 @512"; do
  C=(python tools/conv_bench.py --only "$only" --batch 4 --iters 30)
  echo "plain: $("${C[@]}" 2>/dev/null | grep '^same')"
  echo "rgb:   $("${C[@]}" --rgb 2>/dev/null | grep '^same')"
done
python tools/conv_bench.py --only "rgb " --batch 4 --iters 30 2>/dev/null | grep '^rgb'
