#!/bin/bash
O=gpurun_out/r03f; mkdir -p $O
export PYTHONPATH=$PWD
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{
( timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -k "not two_ranks" 2>&1 | tail -15 )
for s in 5 10; do for o in 3 4 5 1 2; do
  TAG="wave-L2" ITERS=20 T 256 $o $s
  EDHIP_NO_WAVE_L2=1 TAG="old-L2 " ITERS=20 T 256 $o $s
done; done
EDHIP_PRINT_SPILL=1 ITERS=2 timeout 120 python tools/time_k12.py 256 3 10 2>&1 | grep "edhip:" | sort | uniq -c
EDHIP_PRINT_SPILL=1 ITERS=2 timeout 120 python tools/time_k12.py 256 5 5 2>&1 | grep "edhip:" | sort | uniq -c
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
