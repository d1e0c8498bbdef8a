#!/bin/bash
# round 6, experiment b: steps in flight at the headline batch (4 frames per step), same box
Q="python bench.py --no-extras --no-video --no-cpu-baseline"
sum1() { grep '"metric"' | python tools/bench_summary.py | head -1 | cut -c1-60; }
for r in 1 2; do for L in 2 3 4 5; do echo "batch 4 lanes $L: $($Q --batch 4 --lanes $L 2>/dev/null | sum1)"; done; done
