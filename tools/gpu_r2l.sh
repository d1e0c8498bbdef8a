#!/bin/bash
# A/B: style branch / thin-op branch of the frame graph; batched slab loads in the split-K reduce
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 400 python -m pytest tests/test_engine.py tests/test_video.py -m gpu -q -x 2>&1 | tail -2
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 3 --kernels > $O/ab_$name.json 2> $O/ab_$name.err
  python -c "import json; d=json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), 'module', round(d['module_call']['value'],1), d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
for rep in 1 2; do
run l_none$rep VT_STYLE_FORK=0 VT_THIN_FORK=0
run l_style$rep VT_STYLE_FORK=1 VT_THIN_FORK=0
run l_both$rep VT_STYLE_FORK=1 VT_THIN_FORK=1
done
grep "splitk_reduce\|instnorm_stats   \|linear_batch  " $O/ab_l_both2.err | head -5
