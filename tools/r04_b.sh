#!/bin/bash
# round 4: where does hot_grad2_kernel's time go (profiling build, EDHIP_TILE_DBG ablations)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
export PYTHONPATH=$R
cd $R
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
for d in 0 4 8 12; do
  TAG="dbg=$d " EDHIP_TILE_DBG=$d ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
done
TAG="old    " ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
EDHIP_RECORDS=1 TAG="new s10" ITERS=20 timeout 200 python tools/time_k12.py 256 3 10
TAG="old s10" ITERS=20 timeout 200 python tools/time_k12.py 256 3 10
} 2>&1 | grep -v amdgpu.ids > $O/abl.txt
cat $O/abl.txt
