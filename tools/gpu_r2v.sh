#!/bin/bash
# linear_batch 4 columns per wave (HEAD) + RCCL world_size-1 launch of bench.py exactly as the driver's N=1 torchrun form
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_engine.py tests/test_psp.py tests/test_bisenet.py tests/test_ops.py -m gpu -q -x -k "golden or module or linear or style" 2>&1 | tail -2
for rep in 1 2; do
timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 3 --kernels > $O/ab_v$rep.json 2> $O/ab_v$rep.err
python -c "import json; d=json.loads(open('$O/ab_v$rep.json').read().strip().splitlines()[-1]); print('lin4 rep$rep', round(d['value'],1), 'single', round(d['single_stream']['value'],1), round(d['roofline']['kernel_sum_ms_per_frame'],3))"
done
grep "linear_batch  " $O/ab_v2.err | head -4
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-video --no-extras 2>&1 | grep '"metric"' | cut -c1-260
