#!/bin/bash
# round 3, GPU call 2: wave K1 (pipelined planes) + wave K2: tests, timings, ablations, PMC
O=gpurun_out/r03b; mkdir -p $O
export PYTHONPATH=$PWD
( EDHIP_WAVE=3 timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -k "not two_ranks" 2>&1 | tail -60 ) > $O/tests_wave3.txt
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{
for cfg in "0 12800" "3 12800" "3 10240" "3 16384"; do set -- $cfg
  EDHIP_WAVE=$1 EDHIP_WAVE_LDS=$2 TAG="wave=$1 lds=$2" ITERS=30 T 256 3 5; done
EDHIP_WAVE=3 BOXES=0 TAG="wave=3 noboxes" ITERS=30 T 256 3 5
for d in 1 2 3 8 11; do EDHIP_WAVE=3 EDHIP_TILE_DBG=$d TAG="wave=3 dbg=$d" ITERS=20 T 256 3 5; done
for d in 1 2 3; do EDHIP_WAVE=3 EDHIP_WAVE_LDS=10240 EDHIP_TILE_DBG=$d TAG="wave=3 lds=10240 dbg=$d" ITERS=20 T 256 3 5; done
for o in 1 2 4 5; do EDHIP_WAVE=3 TAG="wave=3" ITERS=20 T 256 $o 5; done
EDHIP_WAVE=3 TAG="wave=3" ITERS=20 T 256 3 10
EDHIP_WAVE=3 TAG="wave=3" ITERS=20 T 128 3 5
EDHIP_WAVE=3 EDHIP_PRINT_SPILL=1 TAG="wave=3" ITERS=2 timeout 120 python tools/time_k12.py 256 3 10 2>&1 | grep "edhip:" | sort | uniq -c
EDHIP_WAVE=3 EDHIP_PRINT_SPILL=1 TAG="wave=3" ITERS=2 timeout 120 python tools/time_k12.py 256 3 5 2>&1 | grep "edhip:" | sort | uniq -c
} > $O/sweep.txt 2>&1
EDHIP_WAVE=3 OUTNAME=r03b/pmc_wave bash tools/pmc_hot.sh
cat $O/sweep.txt; tail -5 $O/tests_wave3.txt
