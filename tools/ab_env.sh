#!/bin/bash
# same-box A/B of an environment switch on the headline bench:  tools/ab_env.sh VAR valueA valueB [rounds]
V=$1; A=$2; B=$3; R=${4:-2}
Q="python bench.py --no-extras --no-video --no-cpu-baseline"
for i in $(seq $R); do
  for x in $A $B; do
    echo "$V=$x: $(env $V=$x $Q 2>/dev/null | grep '"metric"' | python tools/bench_summary.py | head -1)"
  done
done
