#!/bin/bash
# round 5, third GPU run: K1 with sampled boxes for general tiles too (two pipelined loops), in-order split reads
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; rm -rf $O; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
T() { timeout 200 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{ TAG="shipped new K1" T 256 3 5; TAG="shipped new K1 s10" T 256 3 10; TAG="shipped o1" T 256 1 5; TAG="shipped o2" T 256 2 5; TAG="shipped 128" T 128 3 5; } > $O/time_ship.txt 2>&1
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
TAG="exp new K1           " T 256 3 5
TAG="exp old K1           " EDHIP_K1_OLD=1 T 256 3 5
TAG="exp new K1, no fast tiles" EDHIP_TILE_DBG=65536 T 256 3 5
TAG="exp new K1 s10       " T 256 3 10
TAG="exp old K1 s10       " EDHIP_K1_OLD=1 T 256 3 10
} > $O/time_exp.txt 2>&1
{ python tools/k1_phases.py 5 3; python tools/k1_phases.py 10 3; } 2>&1 | grep -v amdgpu > $O/phases.txt
cp tools/libedhip_stats.so elasticdeform_amd/libedhip.so
{
for a in "256 3 5" "256 3 10" "256 3 15" "256 3 5 8" "256 1 5" "256 2 5" "256 3 5 5 constant" "256 3 5 5 nearest" "256 3 5 5 reflect" "256 3 5 5 wrap" "128 3 5" "200 3 5" "100 2 5 4 nearest"; do
  timeout 100 python tools/k1_stats.py $a 2>&1 | tail -1
done
} > $O/k1_stats.txt 2>&1
cp /tmp/ship.so elasticdeform_amd/libedhip.so
cat $O/time_ship.txt $O/time_exp.txt $O/phases.txt $O/k1_stats.txt
