#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -15 ) > $O/pytest.txt
cp elasticdeform_amd/libedhip.so /tmp/libedhip_ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
for d in 0 16 32; do
  TAG="dbg=$d " EDHIP_TILE_DBG=$d ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
done
TAG="wg5 24KB" EDHIP_G2_WG5=1 EDHIP_G2_CELLS_KB=24 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="wg4 24KB" EDHIP_G2_CELLS_KB=24 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="wg3 44KB" EDHIP_G2_CELLS_KB=44 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="old    " ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
} 2>&1 | grep -v amdgpu.ids > $O/abl.txt
cp /tmp/libedhip_ship.so elasticdeform_amd/libedhip.so
cat $O/pytest.txt; cat $O/abl.txt
