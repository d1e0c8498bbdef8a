#!/bin/bash
# Same-box A/B experiments, one name per experiment; run on the GPU box through
#   gpurun -- 'bash tools/gpu.sh TAG sh "bash tools/ab.sh NAME [args]"'
# (the round-3 experiments whose switches were retired in round 4 -- AdaIN variants, c64 phases, upblur skew / persist, graph
# forks -- live on in profiles/r03_*.txt)
#   env VAR A B [rounds]   the headline bench (no extras) under VAR=A / VAR=B, alternating
#   pipe                   software-pipelined 256-pixel patch tiles against the per-tap form (VT_PATCH_PIPE=0 / 1), batch 4
#   trunk                  the 32 x 32 trunk of a batch: weight-stationary whole-K kernel (VT_BATCH_EXACT=1) vs 256 x 32 patch tiles
#   enc                    stride-2 encoder convs: tile shapes at batch 4
#   patch                  256x128 patch tiles (no K split) vs the per-image plan, batch 4
#   lanes                  batch x steps in flight on the headline
CB="python tools/conv_bench.py"
Q="python bench.py --no-extras --no-video --no-cpu-baseline"
sum1() { grep '"metric"' | python tools/bench_summary.py | head -1; }
row() { grep -v '^total\|^#\|amdgpu' | tail -1; }
name=$1; shift
case $name in
  env) V=$1; A=$2; B=$3; R=${4:-2}
    for i in $(seq $R); do for x in $A $B; do echo "$V=$x: $(env $V=$x $Q 2>/dev/null | sum1)"; done; done ;;
  pipe)
    for only in "=same 256 @128" "=fus2 512->256 @128" "=enc2.2 512->512 @64" "=fus1 1024->512 @64" "=same 128 @256" "=same 64 @512"; do
      $CB --stream --only "$only" --batch 4 --iters 30 --sweep VT_PATCH_PIPE=0,1 2>/dev/null | grep -v '^total\|amdgpu'
    done ;;
  trunk)
    for only in "=res 512->512 @32" "=fus0 1024->512 @32"; do
      echo "whole-K:   $(VT_BATCH_EXACT=1 $CB --stream --only "$only" --batch 4 --iters 50 2>/dev/null | row)"
      echo "256x32:    $($CB --stream --only "$only" --batch 4 --iters 50 2>/dev/null | row)"
    done ;;
  enc)
    for only in "=enc1.0 128->256 s2" "=enc2.0 256->512 s2" "=enc3.0 512->512 s2"; do for hint in 0 64064 64128 128064 128128; do
      echo "hint $hint: $($CB --only "$only" --batch 4 --iters 50 --hint $hint 2>/dev/null | row)"
    done; done ;;
  patch)
    for only in "=same 256 @128" "=fus2 512->256 @128" "=enc1.2 256->256 @128" "=enc2.2 512->512 @64" "=fus1 1024->512 @64"; do
      echo "per-image plan: $(VT_BATCH_EXACT=1 $CB --stream --only "$only" --batch 4 --iters 30 2>/dev/null | row)"
      for hint in 101256128 101128128; do echo "hint $hint: $($CB --stream --only "$only" --batch 4 --iters 30 --hint $hint 2>/dev/null | row)"; done
    done ;;
  lanes)
    for cfg in "4 2" "4 3" "4 4" "8 2" "8 3" "12 2" "16 1" "16 2"; do set -- $cfg
      echo "batch $1 lanes $2: $($Q --batch $1 --lanes $2 2>/dev/null | sum1)"
    done ;;
  *) echo "unknown experiment $name"; exit 1 ;;
esac
