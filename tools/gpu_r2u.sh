#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/raft_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/raft_bench.txt
timeout 300 python tools/raft_bench.py --window 2 2>&1 | grep -v amdgpu | tee -a gpurun_out/raft_bench.txt
