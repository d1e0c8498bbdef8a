#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { local name=$1; shift
  env "$@" timeout 25 python bench.py --steps 30 --no-cpu-baseline --no-video --op-iters 1 --kernels > gpurun_out/y_$name.json 2> gpurun_out/y_$name.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/y_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'], d['output_checksum']['samples'][:3])"
}
run base A=1
run fullk1 VT_FULLK=1
run fullk2 VT_FULLK=2
