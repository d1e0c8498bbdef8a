#!/bin/bash
# Same-box A/B experiments of round 3 (second half), one name per experiment; run on the GPU box through
#   gpurun -- 'bash tools/gpu.sh TAG sh "bash tools/ab.sh NAME [args]"'
# Outputs are the blocks of profiles/r03_adain_ab.txt and profiles/r03_batch_tiles.txt.
#   env VAR A B [rounds]   the headline bench (no extras) under VAR=A / VAR=B, alternating
#   adain      conv_fullkw_kernel: AdaIN consumer / statistics producer / residual variants + ablations 44-46
#   upblur     conv_upblur tile width / double buffering on the two deepest levels, batch 4
#   tall       conv_upblur tall tiles (24 x 16 quads, 8 waves) vs 12-row tiles, batch 4
#   p8         conv_upblur persistent 8-wave form on the 1024^2 level (kernel, frame, batch 1)
#   top        top levels: 4-wave persistent / uncapped-register conv_upblur, c64 vs the patch kernel
#   enc        stride-2 encoder convs: tile shapes at batch 4
#   patch      256x128 patch tiles (no K split) vs the per-image plan, batch 4
#   lanes      batch x steps in flight on the headline
#   c64        conv3x3_c64_kernel phase ablations, with / without fused ToRGB
#   c32        conv3x3_c32_kernel fused-ToRGB ablations
#   rgb        fused ToRGB vs plain conv vs a separate ToRGB launch
CB="python tools/conv_bench.py"
Q="python bench.py --no-extras --no-video --no-cpu-baseline"
sum1() { grep '"metric"' | python tools/bench_summary.py | head -1; }
row() { grep -v '^total\|^#\|amdgpu' | tail -1; }
name=$1; shift
case $name in
  env) V=$1; A=$2; B=$3; R=${4:-2}
    for i in $(seq $R); do for x in $A $B; do echo "$V=$x: $(env $V=$x $Q 2>/dev/null | sum1)"; done; done ;;
  adain) C=($CB --stream --only "=res 512->512 @32" --batch 4 --iters 50)
    for a in 0 1 2 3 4 5 6 7; do echo "adain=$a: $("${C[@]}" --adain $a 2>/dev/null | row)"; done
    for abl in 44 45; do for a in 1 3; do echo "ABL $abl adain=$a: $(VT_FULLKW_ABLATE=$abl "${C[@]}" --adain $a 2>/dev/null | row)"; done; done
    for a in 2 3; do echo "ABL 46 adain=$a: $(VT_FULLKW_ABLATE=46 "${C[@]}" --adain $a 2>/dev/null | row)"; done ;;
  upblur)
    for only in "=up 512->512 @32->64" "=up 512->256 @64->128"; do for hint in 16 32; do for db in 99 4; do
      echo "hint $hint DB>=$db: $(VT_UPBLUR_DB=$db $CB --upblur --only "$only" --batch 4 --iters 50 --hint $hint 2>/dev/null | row)"
    done; done; done ;;
  tall)
    for only in "=up 512->512 @32->64" "=up 512->256 @64->128" "=up 256->128 @128->256" "=up 128->64 @256->512"; do for t in 0 1; do
      echo "tall>=$t: $(VT_UPBLUR_TALL=$t VT_UPBLUR_DB=99 $CB --upblur --only "$only" --batch 4 --iters 50 --hint 32 2>/dev/null | row)"
    done; done ;;
  p8)
    for b in 1 4; do for x in 0 1; do
      echo "batch $b P8=$x: $(VT_UPBLUR_P8=$x $CB --upblur --batch $b --iters 30 --hint 32 --only "=up 64->32 @512->1024" 2>/dev/null | row)"
    done; done
    bash tools/ab.sh env VT_UPBLUR_P8 0 1 2
    for x in 0 1; do echo "batch1 VT_UPBLUR_P8=$x: $(env VT_UPBLUR_P8=$x $Q --batch 1 2>/dev/null | sum1)"; done ;;
  top) U=($CB --upblur --batch 4 --iters 30 --hint 32)
    for only in "=up 64->32 @512->1024" "=up 128->64 @256->512"; do
      echo "default:    $(VT_UPBLUR_P8=0 "${U[@]}" --only "$only" 2>/dev/null | row)"
      echo "persist:    $(VT_UPBLUR_P8=0 VT_UPBLUR_PERSIST=1 "${U[@]}" --only "$only" 2>/dev/null | row)"
      echo "lb2=0:      $(VT_UPBLUR_P8=0 VT_UPBLUR_LB2=0 "${U[@]}" --only "$only" 2>/dev/null | row)"
    done
    C=($CB --only "=same 64 @512" --batch 4 --iters 30 --rgb)
    echo "c64:        $("${C[@]}" 2>/dev/null | row)"
    echo "patch:      $(VT_C64_KERNEL=0 "${C[@]}" 2>/dev/null | row)" ;;
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
  c64) C=($CB --only "=same 64 @512" --batch 4 --iters 30)
    echo "plain:      $("${C[@]}" 2>/dev/null | row)"
    echo "rgb:        $("${C[@]}" --rgb 2>/dev/null | row)"
    # VT_RGB_ABLATE: 31 no activation stores, 32 no tap loop, 33 no patch loads, 1 no rgb stores, 2 no skip loads, 3 no shuffles,
    # 34 no exchange barrier, 35 no dot products
    for a in 31 32 33 1 2 3 34 35; do echo "rgb ABL $a: $(VT_RGB_ABLATE=$a "${C[@]}" --rgb 2>/dev/null | row)"; done
    for a in 31 32 33; do echo "plain ABL $a: $(VT_RGB_ABLATE=$a "${C[@]}" 2>/dev/null | row)"; done
    echo "rows16 rgb: $(VT_C64_ROWS=16 "${C[@]}" --rgb 2>/dev/null | row)"
    echo "pipe1 rgb:  $(VT_C64_PIPE=1 "${C[@]}" --rgb 2>/dev/null | row)" ;;
  c32) C=($CB --only "=same 32 @1024" --batch 4 --iters 30)
    echo "plain:   $("${C[@]}" 2>/dev/null | row)"
    echo "rgb:     $("${C[@]}" --rgb 2>/dev/null | row)"
    for a in 1 2 3; do echo "rgb ABL $a: $(VT_RGB_ABLATE=$a "${C[@]}" --rgb 2>/dev/null | row)"; done ;;
  rgb)
    for only in "=same 128 @256" "=same 64 @512"; do
      echo "plain: $($CB --only "$only" --batch 4 --iters 30 2>/dev/null | row)"
      echo "rgb:   $($CB --only "$only" --batch 4 --iters 30 --rgb 2>/dev/null | row)"
    done
    $CB --only "rgb " --batch 4 --iters 30 2>/dev/null | grep '^rgb' ;;
  pipe)   # round 4: software-pipelined 256-pixel patch tiles (VT_PATCH_PIPE: 0 per-tap form, 1 pinned 1:1, 2 unpinned, 3 pinned 1:2)
    for only in "=same 256 @128" "=fus2 512->256 @128" "=enc2.2 512->512 @64" "=fus1 1024->512 @64" "=same 128 @256"; do
      $CB --stream --only "$only" --batch 4 --iters 30 --sweep ${PIPE_SWEEP:-VT_PATCH_PIPE=0,1,2,3} 2>/dev/null | grep -v '^total\|amdgpu'
    done
    $CB --dtype fp32 --stream --only "=same 256 @128" --batch 4 --iters 10 --sweep VT_PATCH_PIPE=0,1 2>/dev/null | grep -v '^total\|amdgpu' ;;
  *) echo "unknown experiment $name"; exit 1 ;;
esac
