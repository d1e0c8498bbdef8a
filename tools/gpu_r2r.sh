#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for rep in 1 2; do
for L in 2 3 4 5 6; do
timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 1 --lanes $L --min-seconds 0.6 > $O/ab_lanes$L.json 2> $O/ab_lanes$L.err
python -c "import json; d=json.loads(open('$O/ab_lanes$L.json').read().strip().splitlines()[-1]); print('lanes $L', round(d['value'],1), 'single', round(d['single_stream']['value'],1))"
done
done
