#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -k "whole_k or adain" 2>&1 | tail -2
for rep in 1 2; do
for dp in 6 9; do
VT_FULLK_DEPTH=$dp timeout 60 python tools/conv_bench.py --only "res 512" --iters 100 --stream 2>/dev/null | grep -v total | awk -v t="DEPTH=$dp" '{print t, $1,$2,$3, $(NF-5), $(NF-4)}'
VT_FULLK_DEPTH=$dp timeout 60 python tools/conv_bench.py --only "fus0" --iters 100 --stream 2>/dev/null | grep -v total | awk -v t="DEPTH=$dp" '{print t, $1,$2,$3, $(NF-5), $(NF-4)}'
VT_FULLK_DEPTH=$dp timeout 60 python tools/conv_bench.py --only "enc2.2" --iters 50 --stream 2>/dev/null | grep -v total | awk -v t="DEPTH=$dp" '{print t, $1,$2,$3, $(NF-5), $(NF-4)}'
done
done
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 3 > $O/ab_$name.json 2> $O/ab_$name.err
  python -c "import json; d=json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
run d6 VT_FULLK_DEPTH=6
run d9 VT_FULLK_DEPTH=9
run d6b VT_FULLK_DEPTH=6
run d9b VT_FULLK_DEPTH=9
