# Same-box A/B of two builds of the library (this tree against the one of commit 432b42a, built in a git worktree and copied to
# gpurun_ab/): the dilation-4 trunk conv and the 1024^2 conv + ToRGB, three alternations.  -> profiles/r06_ab_flat8_c32.txt
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in "" "--lib gpurun_ab/libvt_432b42a.so"; do
    echo "## lib=${L:-new} rep $i"
    python tools/conv_bench.py --only "modres 512" --batch 4 --stream $L 2>&1 | grep -E "^modres"
    python tools/conv_bench.py --only "same 32 @1024" --batch 4 --rgb $L 2>&1 | grep -E "^same"
  done
done
