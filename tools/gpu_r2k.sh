#!/bin/bash
# fullk stream experiments: size scaling (how many CUs stream at once), cache policy of the weight loads, sharing degree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r2k.txt; : > $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> $O
CB="python tools/conv_bench.py --stream --hint 400000000 --iters 200"
for rep in 1 2; do
echo "== base rep $rep" >> $O; timeout 120 $CB --only "res 512" 2>&1 | grep -v "^total" >> $O
for a in 1 2 3; do echo "== aux $a" >> $O; VT_FULLK_AUX=$a timeout 120 $CB --only "=res 512->512 @32" 2>&1 | grep -v "^total" >> $O; done
for m in 23 24 21 22; do echo "== ablate $m" >> $O; VT_RGB_ABLATE=$m timeout 120 $CB --only "=res 512->512 @32" 2>&1 | grep -v "^total" >> $O; done
echo "== depth 9" >> $O; VT_FULLK_DEPTH=9 timeout 120 $CB --only "=res 512->512 @32" 2>&1 | grep -v "^total" >> $O
done
echo "== fus/b4" >> $O; timeout 120 $CB --only "fus0" >> $O 2>&1; timeout 120 $CB --only "=res 512->512 @32" --batch 4 >> $O 2>&1
cat $O
