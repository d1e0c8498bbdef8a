#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_bisenet.py tests/test_video.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/pytest_bisenet.log; tail -4 gpurun_out/pytest_bisenet.log
timeout 300 python tools/bisenet_bench.py > gpurun_out/bisenet_bench.log 2>&1; grep -v "^W\|^E" gpurun_out/bisenet_bench.log | tail -12
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bis -o b -- python $GRAFT_REPO_ROOT/tools/bisenet_bench.py --no-cpu --no-video --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/prof_bis.log 2>&1)
python tools/rocpd_stats.py $(find gpurun_out/prof_bis -name "*.db" | head -1) > gpurun_out/prof_bis_stats.txt 2>&1; rm -rf gpurun_out/prof_bis
cut -c1-150 gpurun_out/prof_bis_stats.txt | head -24
