#!/bin/bash
for b in 1 4; do
U=(python tools/conv_bench.py --upblur --batch $b --iters 30 --hint 32 --only "=up 64->32 @512->1024")
echo "batch $b off: $(VT_UPBLUR_P8=0 "${U[@]}" 2>/dev/null | grep '^up')"
echo "batch $b on:  $(VT_UPBLUR_P8=1 "${U[@]}" 2>/dev/null | grep '^up')"
done
bash tools/ab_env.sh VT_UPBLUR_P8 0 1 2
Q="python bench.py --no-extras --no-video --no-cpu-baseline --batch 1"
for x in 0 1; do echo "batch1 VT_UPBLUR_P8=$x: $(env VT_UPBLUR_P8=$x $Q 2>/dev/null | grep '"metric"' | python tools/bench_summary.py | head -1)"; done
