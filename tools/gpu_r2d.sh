#!/bin/bash
# Round-2 GPU pass D: conv_transpose+blur kernel -- parity on hardware, micro-benchmark vs the polyphase form, frame A/B.
TAG=${1:-r2d}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -k "transpose_blur or whole_k or styled" 2>&1 | tail -5 > $O/pytest_ops_$TAG.log; tail -2 $O/pytest_ops_$TAG.log
timeout 400 python -m pytest tests/test_engine.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_eng_$TAG.log; tail -3 $O/pytest_eng_$TAG.log
( timeout 100 python tools/conv_bench.py --only "up " --iters 50 2>/dev/null | grep -v total
  timeout 100 python tools/conv_bench.py --only "up " --iters 50 --upblur 2>/dev/null | grep -v total | sed 's/^/   UPBLUR /'
  timeout 100 python tools/conv_bench.py --only "up " --iters 50 --upblur --hint 16 2>/dev/null | grep -v total | sed 's/^/   UPBLUR16 /'
  timeout 100 python tools/conv_bench.py --only "up " --iters 20 --upblur --batch 4 2>/dev/null | grep -v total | sed 's/^/   UPBLUR B4 /'
  timeout 100 python tools/conv_bench.py --only "up " --iters 20 --batch 4 2>/dev/null | grep -v total | sed 's/^/   B4 /' ) > $O/convbench_$TAG.txt 2>&1
cat $O/convbench_$TAG.txt
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --op-iters 3 --kernels > $O/ab_${TAG}_$name.json 2> $O/ab_${TAG}_$name.err
  python -c "import json; d=json.loads(open('$O/ab_${TAG}_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), 'module', round(d['module_call']['value'],1) if d.get('module_call') else None, 'b4', round(d['batch4']['value'],1) if d.get('batch4') else None, 'c3', round(d['config3']['value'],1) if d.get('config3') else None, d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
run upblur A=1
run polyphase VT_UPBLUR=0
run upblur2 A=1
grep -v "^W\|^E\|amdgpu.ids" $O/ab_${TAG}_upblur.err | head -130 > $O/kernels_$TAG.txt
head -24 $O/kernels_$TAG.txt | cut -c1-140
grep "upblur" $O/kernels_$TAG.txt | tail -6
