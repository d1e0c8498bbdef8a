#!/bin/bash
# First GPU call of the next round: parity tier, the A/B switches prepared without GPU time,
# the per-layer plan search with three frames in flight, and the full profile pass.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_next_round.sh r02a'
TAG=${1:-r02a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu_$TAG.log; tail -3 gpurun_out/pytest_gpu_$TAG.log
run() { local name=$1; shift
  env "$@" timeout 60 python bench.py --steps 40 --no-cpu-baseline --no-video --op-iters 1 > gpurun_out/ab_${TAG}_$name.json 2> gpurun_out/ab_${TAG}_$name.err
  python -c "import json; d=json.loads(open('gpurun_out/ab_${TAG}_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'])"
}
run base A=1
run small_lds VT_SMALL_LDS=1
run occ2 VT_PATCH_OCC2=1
run small_lds_occ2 VT_SMALL_LDS=1 VT_PATCH_OCC2=1
run wg128 VT_SPLITK_WGS=128
run small_lds_l4 VT_SMALL_LDS=1 VT_PATCH_OCC2=1 VT_BENCH_LANES=4
run hwq8_l4 GPU_MAX_HW_QUEUES=8 VT_BENCH_LANES=4
run hwq8_l6 GPU_MAX_HW_QUEUES=8 VT_BENCH_LANES=6
run base2 A=1
timeout 700 python tools/plan_sweep.py --lanes 3 --top 14 --budget-s 600 --out gpurun_out/tile_hints_$TAG.json > gpurun_out/plan_sweep_$TAG.log 2>&1; grep -v "^W\|^E" gpurun_out/plan_sweep_$TAG.log | tail -25
[ -s gpurun_out/tile_hints_$TAG.json ] && run hinted VT_TILE_HINTS=$GRAFT_REPO_ROOT/gpurun_out/tile_hints_$TAG.json
bash tools/gpu_final.sh $TAG
