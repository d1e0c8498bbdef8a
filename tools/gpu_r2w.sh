#!/bin/bash
# A/B: Fusion gate (AdaIN affine + mask conv + pack) in one launch vs three
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -k "fusion_gate or thin" 2>&1 | tail -1
timeout 400 python -m pytest tests/test_engine.py -m gpu -q -x -k "golden or full_size_fp32 or config3 or properties" 2>&1 | tail -1
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 3 --kernels > $O/ab_$name.json 2> $O/ab_$name.err
  python -c "import json; d=json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
for rep in 1 2; do
run w_three$rep VT_FUSE_GATE=0
run w_gate$rep VT_DUMMY=1
done
grep "fusion_gate" $O/ab_w_gate2.err | head -6
