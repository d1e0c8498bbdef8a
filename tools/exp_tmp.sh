CB="python tools/conv_bench.py"
$CB --stream --only "=fus1 1024->512 @64" --batch 4 --iters 30 --sweep VT_PATCH_PIPE=1,4,9,a 2>&1 | grep -v 'amdgpu\|^total'
