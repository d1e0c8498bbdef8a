#!/bin/bash
# A/B: thin-output conv kernel (one launch, scatter form) vs tile kernels + split-K; patch tiles for the top same-res convs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -k "thin or conv_shapes or in_launch" 2>&1 | tail -2
timeout 400 python -m pytest tests/test_engine.py -m gpu -q -x -k "golden or full_size_fp32 or config3" 2>&1 | tail -2
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 3 --kernels > $O/ab_$name.json 2> $O/ab_$name.err
  python -c "import json; d=json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
for rep in 1 2; do
run n_old$rep VT_THIN_KERNEL=0
run n_thin$rep VT_DUMMY=1
done
grep "conv_thin" $O/ab_n_thin2.err | head -14
CB="python tools/conv_bench.py --iters 100 --rgb"
for h in 0 100256064 100128064; do echo "hint $h"; timeout 60 $CB --only "same 64 @512" --hint $h 2>&1 | grep -v "^total\|amdgpu"; done
for h in 0 100256128 100128128 100128064; do echo "hint $h"; timeout 60 $CB --only "same 128 @256" --hint $h 2>&1 | grep -v "^total\|amdgpu"; done
