#!/bin/bash
# Round-2 GPU pass A: whole-K trunk kernel -- parity on hardware, micro-benchmark against the slab path,
# whole-frame A/B.  usage: gpurun --timeout 900 -- 'bash tools/gpu_r2a.sh r2a'
TAG=${1:-r2a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -k "whole_k or mfma or conv_shapes" 2>&1 | tail -5 > $O/pytest_ops_$TAG.log; tail -2 $O/pytest_ops_$TAG.log
timeout 300 python -m pytest tests/test_engine.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_eng_$TAG.log; tail -3 $O/pytest_eng_$TAG.log
for only in "res 512" "modres" "fus0" "same 512 @64" "enc2.2" "fus1"; do
  timeout 60 python tools/conv_bench.py --only "$only" --iters 50 2>/dev/null | grep -v total
  timeout 60 python tools/conv_bench.py --only "$only" --iters 50 --stream --hint 400000000 2>/dev/null | grep -v total | sed 's/^/   STREAM /'
  timeout 60 python tools/conv_bench.py --only "$only" --iters 50 --stream --hint 400000000 --batch 4 2>/dev/null | grep -v total | sed 's/^/   STREAM B4 /'
  timeout 60 python tools/conv_bench.py --only "$only" --iters 50 --batch 4 2>/dev/null | grep -v total | sed 's/^/   B4 /'
done > $O/convbench_$TAG.txt 2>&1
cat $O/convbench_$TAG.txt
run() { local name=$1; shift
  env "$@" timeout 90 python bench.py --steps 100 --no-cpu-baseline --no-video --op-iters 3 --kernels > $O/ab_${TAG}_$name.json 2> $O/ab_${TAG}_$name.err
  python -c "import json; d=json.loads(open('$O/ab_${TAG}_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3))"
}
run fullk A=1
run nofullk VT_FULLK_KERNEL=0
run fullk_trunk_only VT_FULLK_MAX_WGS=256
run small_lds VT_SMALL_LDS=1
run fullk2 A=1
grep -v "^W\|^E\|amdgpu.ids" $O/ab_${TAG}_fullk.err | head -120 > $O/kernels_$TAG.txt
