#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03hint; mkdir -p $O
cp elasticdeform_amd/libedhip.so /tmp/ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{
for o in 4 5; do for s in 5 10; do
TAG="feedback o$o s$s" ITERS=20 T 256 $o $s
EDHIP_NO_SPILL_HINT=1 TAG="standard o$o s$s" ITERS=20 T 256 $o $s
done; done
} > $O/ab45.txt 2>&1
timeout 800 python tools/cmp_levels.py 2>&1 | grep -v amdgpu > $O/cmp2.txt
cp /tmp/ship.so elasticdeform_amd/libedhip.so
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest2.txt
( timeout 400 python tests/fuzz/fuzz_hot.py 84 150 2>&1 | tail -4 ) >> $O/pytest2.txt
cat $O/ab45.txt; grep -v " 0 of" $O/cmp2.txt; echo ---; cat $O/pytest2.txt
