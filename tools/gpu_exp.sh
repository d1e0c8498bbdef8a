#!/bin/bash
# experiment: split-K workgroup target under 3 frames in flight; video driver with separate copy streams
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for WG in 64 128 256; do for L in 1 3; do
  VT_SPLITK_WGS=$WG timeout 200 python bench.py --lanes $L --no-cpu-baseline --no-video --op-iters 1 > gpurun_out/sk_${WG}_$L.json 2> gpurun_out/sk_${WG}_$L.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/sk_${WG}_$L.json').read().strip().splitlines()[-1]); print('WG',$WG,'L',$L, round(d['value'],1), round(d['ms_per_step'],3))"
done; done
timeout 300 python tools/video_bench.py --frames 192 > gpurun_out/video_bench.log 2>&1; grep "^batch [0-9]" gpurun_out/video_bench.log | cut -c1-80
timeout 100 python -m pytest tests/test_video.py -m gpu -q -x 2>&1 | tail -2
