#!/bin/bash
# A/B: persistent 64->64 kernel (512^2 level) vs the patch-resident tile kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -k "c64 or fused_torgb or c32 or thin or in_launch" 2>&1 | tail -2
timeout 400 python -m pytest tests/test_engine.py -m gpu -q -x -k "golden or full_size_fp32 or config3" 2>&1 | tail -2
CB="python tools/conv_bench.py --iters 100"
for v in 0 1; do echo "c64=$v"; VT_C64_KERNEL=$v timeout 60 $CB --only "same 64 @512" --rgb 2>&1 | grep -v "^total\|amdgpu"; VT_C64_KERNEL=$v timeout 60 $CB --only "same 64 @512" 2>&1 | grep -v "^total\|amdgpu"; done
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 3 --kernels > $O/ab_$name.json 2> $O/ab_$name.err
  python -c "import json; d=json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
for rep in 1 2; do
run p_old$rep VT_C64_KERNEL=0
run p_c64$rep VT_DUMMY=1
done
grep "c64" $O/ab_p_c642.err | head -4
