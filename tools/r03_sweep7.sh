#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
export PYTHONPATH=$PWD
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{
for s in 5 10; do for o in 4 5; do
  TAG="oldL1+waveL2" ITERS=20 T 256 $o $s
  EDHIP_WAVE=3 TAG="waveL1+waveL2" ITERS=20 T 256 $o $s
done; done
TAG="oldL1+waveL2" ITERS=30 T 256 3 5
EDHIP_NO_WAVE_L2=1 TAG="oldL1 oldL2 " ITERS=30 T 256 3 5
EDHIP_WAVE=3 EDHIP_PRINT_SPILL=1 ITERS=2 timeout 120 python tools/time_k12.py 256 5 5 2>&1 | grep "edhip:" | sort | uniq -c
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
