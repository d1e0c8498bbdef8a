CB="python tools/conv_bench.py"
B=vtoonify_amd/lib/ab/libvtoonify_amd_before.so
for r in 1 2; do
echo "before: $($CB --upblur --only "=up 64->32 @512->1024" --batch 4 --iters 30 --hint 32 --lib $B 2>/dev/null | grep -v '^total\|amdgpu' | tail -1)"
echo "after:  $($CB --upblur --only "=up 64->32 @512->1024" --batch 4 --iters 30 --hint 32 2>/dev/null | grep -v '^total\|amdgpu' | tail -1)"
done
