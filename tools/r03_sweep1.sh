#!/bin/bash
# round 3, GPU call 1: correctness of the wave-per-tile K1 + first timings
O=gpurun_out/r03a; mkdir -p $O
export PYTHONPATH=$PWD
( EDHIP_WAVE=1 timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -x -k "not two_ranks" 2>&1 | tail -40 ) > $O/tests_wave1.txt
for cfg in "0 12800 4" "1 12800 4" "1 10240 4" "1 16384 4" "1 12800 2" "1 12800 8" "1 12800 1"; do
  set -- $cfg
  EDHIP_WAVE=$1 EDHIP_WAVE_LDS=$2 EDHIP_WAVE_STRIP=$3 TAG="wave=$1 lds=$2 strip=$3" ITERS=30 timeout 120 python tools/time_k12.py 256 3 5 2>&1 | tail -1
done > $O/k1_sweep.txt
for o in 1 2 4 5; do for w in 0 1; do
  EDHIP_WAVE=$w TAG="wave=$w" ITERS=20 timeout 120 python tools/time_k12.py 256 $o 5 2>&1 | tail -1
done; done >> $O/k1_sweep.txt
for w in 0 1; do
  EDHIP_WAVE=$w TAG="wave=$w" ITERS=20 timeout 120 python tools/time_k12.py 256 3 10 2>&1 | tail -1
  EDHIP_WAVE=$w TAG="wave=$w" ITERS=20 timeout 120 python tools/time_k12.py 128 3 5 2>&1 | tail -1
  EDHIP_WAVE=$w EDHIP_PRINT_SPILL=1 TAG="wave=$w" ITERS=2 timeout 120 python tools/time_k12.py 256 3 10 2>&1 | grep "edhip:" | sort | uniq -c
done >> $O/k1_sweep.txt
cat $O/k1_sweep.txt
