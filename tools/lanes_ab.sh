#!/bin/bash
# frames in flight x batch: headline sensitivity (same box)
Q="python bench.py --no-extras --no-video --no-cpu-baseline"
for cfg in "4 3" "8 2" "8 3" "12 1" "12 2" "16 1" "16 2"; do set -- $cfg
  echo "batch $1 lanes $2: $($Q --batch $1 --lanes $2 2>/dev/null | grep '"metric"' | python tools/bench_summary.py | head -1)"
done
