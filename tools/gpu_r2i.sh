#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for rep in 1 2; do
for ab in 0 21 22; do
VT_RGB_ABLATE=$ab timeout 60 python tools/conv_bench.py --only "res 512" --iters 100 --stream 2>/dev/null | grep -v total | awk -v t="ABLATE=$ab" '{print t, $1,$2,$3, $(NF-5), $(NF-4)}'
done
done
run() { local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-video --no-extras --op-iters 3 --kernels > $O/ab_$name.json 2> $O/ab_$name.err
  python -c "import json; d=json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'single', round(d['single_stream']['value'],1), d['output_checksum']['mean_abs'], round(d['roofline']['kernel_sum_ms_per_frame'],3), d['timed_blocks'])"
}
run base A=1
run notorgb VT_FUSE_TORGB=0
run base2 A=1
run notorgb2 VT_FUSE_TORGB=0
grep "conv_patch_kernel<bf16,256x64>\|c32\|128x16> m=  262144\|128x16> m= 1048576\|conv_igemm" $O/ab_base.err | tail -8
grep "conv_patch_kernel<bf16,256x64>\|c32\|128x16> m=  262144\|128x16> m= 1048576\|conv_igemm" $O/ab_notorgb.err | tail -10
