#!/bin/bash
TAG=${1:-r2c}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
rm -f $O/parity_metrics.jsonl
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest_gpu_$TAG.log 2>&1; tail -6 $O/pytest_gpu_$TAG.log
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --op-iters 3 --kernels > $O/ab_${TAG}_$name.json 2> $O/ab_${TAG}_$name.err
  python -c "import json; d=json.loads(open('$O/ab_${TAG}_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), 'module', round(d['module_call']['value'],1) if d.get('module_call') else None, 'b4', round(d['batch4']['value'],1) if d.get('batch4') else None, 'c3', round(d['config3']['value'],1) if d.get('config3') else None, d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
run adain A=1
run noadain VT_FUSE_ADAIN=0
run adain2 A=1
run noadain2 VT_FUSE_ADAIN=0
grep -v "^W\|^E\|amdgpu.ids" $O/ab_${TAG}_adain.err | head -130 > $O/kernels_$TAG.txt
head -24 $O/kernels_$TAG.txt | cut -c1-140
grep "conv_fullk_kernel<bf16,64x32>  m=    1024" $O/ab_${TAG}_adain.err | awk '{print $8}' | tr '\n' ' '
