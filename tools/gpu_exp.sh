#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { local name=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --no-video --op-iters 1 > gpurun_out/x_$name.json 2> gpurun_out/x_$name.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/x_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3), d['single_stream']['value'])"
}
run base A=1
run wg128 VT_SPLITK_WGS=128
run wg64 VT_SPLITK_WGS=64
run wg32 VT_SPLITK_WGS=32
run base2 A=1
run wg128b VT_SPLITK_WGS=128
