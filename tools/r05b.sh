#!/bin/bash
# round 5, second GPU run: K1 with separate general / fast loops and the grouped reverse-order gather
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; rm -rf $O; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or cfg or ragged or wide or stale or properties or reduced or batch or crop" > $O/tests.txt 2>&1; tail -4 $O/tests.txt
T() { timeout 200 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{ TAG="shipped new K1" T 256 3 5; TAG="shipped new K1 s10" T 256 3 10; TAG="shipped o1" T 256 1 5; TAG="shipped o2" T 256 2 5; TAG="shipped 128" T 128 3 5; } > $O/time_ship.txt 2>&1
OUTNAME=r05b/pmc bash tools/pmc_hot.sh
cd $R
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
TAG="exp new K1 split2    " T 256 3 5
TAG="exp new K1 split3    " EDHIP_K1_SPLIT=3 T 256 3 5
TAG="exp new K1 split1    " EDHIP_K1_SPLIT=1 T 256 3 5
TAG="exp new K1 split0    " EDHIP_K1_SPLIT=0 T 256 3 5
TAG="exp old K1           " EDHIP_K1_OLD=1 T 256 3 5
TAG="exp new K1 3 WG/CU (52 KB)" EDHIP_HOT_FWD_KB=52 T 256 3 5
TAG="exp new K1 2 WG/CU (64 KB)" EDHIP_HOT_FWD_KB=64 T 256 3 5
TAG="exp new K1, no fast tiles" EDHIP_TILE_DBG=65536 T 256 3 5
TAG="exp new K1 strip 2   " EDHIP_STRIP=2 T 256 3 5
TAG="exp new K1 strip 1   " EDHIP_STRIP=1 T 256 3 5
} > $O/time_exp.txt 2>&1
{ python tools/k1_phases.py 5 3; python tools/k1_phases.py 5 1; python tools/k1_phases.py 10 3; } 2>&1 | grep -v amdgpu > $O/phases.txt
cp /tmp/ship.so elasticdeform_amd/libedhip.so
cat $O/time_ship.txt $O/time_exp.txt $O/phases.txt; grep -A9 "k1_fwd_kernel" $O/pmc/summary.txt | head -40
