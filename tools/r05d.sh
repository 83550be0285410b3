#!/bin/bash
# round 5, fourth GPU run: K1 final candidates
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; rm -rf $O; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
T() { timeout 200 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{ TAG="shipped new K1" T 256 3 5; TAG="shipped new K1 s10" T 256 3 10; TAG="shipped o1" T 256 1 5; TAG="shipped o2" T 256 2 5; TAG="shipped 128" T 128 3 5; } > $O/time_ship.txt 2>&1
OUTNAME=r05d/pmc bash tools/pmc_k1.sh
cd $R
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
TAG="exp new K1           " T 256 3 5
TAG="exp old K1           " EDHIP_K1_OLD=1 T 256 3 5
TAG="exp new K1, no fast tiles" EDHIP_TILE_DBG=65536 T 256 3 5
TAG="ablation: no gather (131072)" EDHIP_TILE_DBG=131072 T 256 3 5
TAG="ablation: no staging (262144)" EDHIP_TILE_DBG=262144 T 256 3 5
TAG="ablation: no displacement (524288)" EDHIP_TILE_DBG=524288 T 256 3 5
TAG="ablation: no stores (1048576)" EDHIP_TILE_DBG=1048576 T 256 3 5
TAG="ablation: no gather, no staging" EDHIP_TILE_DBG=393216 T 256 3 5
TAG="ablation: no gather, no displacement" EDHIP_TILE_DBG=655360 T 256 3 5
TAG="ablation: no gather, staging, displacement, stores" EDHIP_TILE_DBG=1966080 T 256 3 5
TAG="exp new K1 strip 2   " EDHIP_STRIP=2 T 256 3 5
} > $O/time_exp.txt 2>&1
{ python tools/k1_phases.py 5 3; } 2>&1 | grep -v amdgpu > $O/phases.txt
cp tools/libedhip_exp_s8.so elasticdeform_amd/libedhip.so
{ TAG="exp s8 lib strip 4" T 256 3 5; TAG="exp s8 lib strip 8" EDHIP_STRIP=8 T 256 3 5; TAG="exp s8 lib strip 8 s10" EDHIP_STRIP=8 T 256 3 10;  TAG="exp s8 lib strip 8 128" EDHIP_STRIP=8 T 128 3 5; } >> $O/time_exp.txt 2>&1
cp tools/libedhip_stats.so elasticdeform_amd/libedhip.so
{
for a in "256 3 5" "256 3 10" "256 2 5" "256 3 5 5 constant" "256 3 5 5 nearest" "256 3 5 5 reflect" "256 3 5 5 wrap" "100 2 5 4 nearest" "90 2 3 4 reflect"; do
  timeout 100 python tools/k1_stats.py $a 2>&1 | tail -1
done
} > $O/k1_stats.txt 2>&1
cp /tmp/ship.so elasticdeform_amd/libedhip.so
cat $O/time_ship.txt $O/time_exp.txt $O/phases.txt $O/k1_stats.txt; grep -A9 "k1_fwd_kernel" $O/pmc/summary.txt | head -60
