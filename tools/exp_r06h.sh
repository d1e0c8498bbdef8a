# Same-box A/B of the flat up-sampling tiles before / after the LDS swizzle fix (gpurun_ab/libvt_head.so = commit 78f2937 built in a
# worktree): the three deep up-sampling convs at 4 and 1 frames, three alternations.  -> profiles/r06_upflat.txt section 6
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in "--lib gpurun_ab/libvt_head.so" ""; do
    echo "## lib=${L:-this tree} rep $i"
    python tools/conv_bench.py --upblur --only "@" --batch 4 $L 2>&1 | grep -E "^up (512|256)"
    python tools/conv_bench.py --upblur --only "@" --batch 1 $L 2>&1 | grep -E "^up (512|256)"
  done
done
