#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
run() { local name=$1; shift
  env "$@" timeout 100 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 1 --min-seconds 0.6 > $O/ab_$name.json 2> $O/ab_$name.err
  python -c "import json; d=json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1))"
}
run x_def1 VT_DUMMY=1
run x_wgs128a VT_SPLITK_WGS=128
run x_def2 VT_DUMMY=1
run x_wgs128b VT_SPLITK_WGS=128
