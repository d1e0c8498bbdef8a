#!/bin/bash
# round 6, experiment a: one barrier per chunk (conv_patchc_kernel) against the tap-granular pipeline on the trunk convs, same box
#   gpurun -- 'bash tools/gpu.sh r06c sh "bash tools/exp_r06a.sh"'
CB="python tools/conv_bench.py"
for only in "=res 512->512 @32" "=fus0 1024->512 @32"; do
  $CB --stream --only "$only" --batch 4 --iters 50 --sweep VT_PATCH_PIPE=1,2 2>/dev/null | grep -v '^total\|amdgpu'
  $CB --stream --only "$only" --batch 4 --iters 50 --sweep VT_PATCH_PIPE=1,2 2>/dev/null | grep -v '^total\|amdgpu'
done
echo "--- kernel names (rocprofv3) of the default plan"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_exp -o t -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --stream --only "=res 512->512 @32" --batch 4 --iters 20 > /tmp/prof_exp.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof_exp -name "*.db" | head -1) 2>&1 | head -8 | cut -c1-180
