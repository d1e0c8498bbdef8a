#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for Q in 8 2; do
for L in 3 4 6; do
GPU_MAX_HW_QUEUES=$Q timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 1 --lanes $L --min-seconds 0.6 > $O/ab_q${Q}_lanes$L.json 2> $O/ab_q${Q}_lanes$L.err
python -c "import json; d=json.loads(open('$O/ab_q${Q}_lanes$L.json').read().strip().splitlines()[-1]); print('hwq $Q lanes $L', round(d['value'],1), 'single', round(d['single_stream']['value'],1))"
done
done
timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 1 --lanes 3 --min-seconds 0.6 > $O/ab_qdef.json 2>/dev/null
python -c "import json; d=json.loads(open('$O/ab_qdef.json').read().strip().splitlines()[-1]); print('default lanes 3', round(d['value'],1))"
