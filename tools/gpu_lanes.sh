#!/bin/bash
# A/B: frames in flight per GPU (HIP streams with their own plan buffers) x frames per step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_engine.py tests/test_video.py -m gpu -q -x -k "lanes or flight or video" 2>&1 | tail -5 > gpurun_out/pytest_lanes.log
tail -3 gpurun_out/pytest_lanes.log
for B in 1 2 4; do for L in 1 2 3; do
  timeout 200 python bench.py --batch $B --lanes $L --no-cpu-baseline --no-video --op-iters 1 > gpurun_out/lanes_${B}_$L.json 2> gpurun_out/lanes_${B}_$L.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/lanes_${B}_$L.json').read().strip().splitlines()[-1]); print('B',$B,'L',$L, round(d['value'],1), round(d['ms_per_step'],3))"
done; done
timeout 300 python tools/video_bench.py --frames 192 > gpurun_out/video_bench.log 2>&1; grep "batch" gpurun_out/video_bench.log
