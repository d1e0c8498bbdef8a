# Timing A/B of the dominant class with 32x32x16 MFMAs (profiles/r06_mfma32_timing.txt).  gpurun_ab/libvt_mfma32.so = this tree built
# with mma_all() of conv_patch_persist.hpp issuing 8 v_mfma_f32_32x32x16_bf16 per half-step on the registers of its 16
# v_mfma_f32_16x16x32_bf16 (accumulators copied into the 16x16 layout before the epilogue: wrong results, same instruction mix).
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for L in "" "--lib gpurun_ab/libvt_mfma32.so"; do
    echo "## lib=${L:-this tree} rep $i"
    python tools/conv_bench.py --only "@128" --batch 4 $L 2>&1 | grep -E "^(same|fus)"
    python tools/conv_bench.py --only "@256" --batch 4 $L 2>&1 | grep -E "^(same|fus)"
  done
done
