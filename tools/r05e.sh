#!/bin/bash
# round 5: K2 lane map / pitch experiments (profiling build)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; rm -rf $O; mkdir -p $O; cd $R; export PYTHONPATH=$R
T() { timeout 200 python tools/time_k12.py "$@" 2>&1 | tail -1; }
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
TAG="K2 shipped map                " T 256 3 5
TAG="K2 rows16                     " EDHIP_TILE_DBG=2097152 T 256 3 5
TAG="K2 pitch 32                   " EDHIP_TILE_DBG=4194304 T 256 3 5
TAG="K2 rows16 + pitch 32          " EDHIP_TILE_DBG=6291456 T 256 3 5
TAG="K2 rows16 + pitch 32, 32 KB cells" EDHIP_GRAD_BOX_KB=32 EDHIP_TILE_DBG=6291456 T 256 3 5
TAG="K2 shipped, 32 KB cells       " EDHIP_GRAD_BOX_KB=32 T 256 3 5
TAG="K2 rows16 + pitch 32 s10      " EDHIP_TILE_DBG=6291456 T 256 3 10
TAG="K2 shipped s10                " T 256 3 10
TAG="K2 rows16 + pitch 32 o1       " EDHIP_TILE_DBG=6291456 T 256 1 5
TAG="K2 shipped o1                 " T 256 1 5
TAG="K2 rows16 + pitch 32 128      " EDHIP_TILE_DBG=6291456 T 128 3 5
TAG="K2 shipped 128                " T 128 3 5
TAG="K2 rows16+p32 no flush (64)   " EDHIP_TILE_DBG=6291520 T 256 3 5
TAG="K2 shipped no flush (64)      " EDHIP_TILE_DBG=64 T 256 3 5
TAG="K2 shipped no scatter (128)   " EDHIP_TILE_DBG=128 T 256 3 5
} > $O/time_k2.txt 2>&1
cp /tmp/ship.so elasticdeform_amd/libedhip.so
cat $O/time_k2.txt
