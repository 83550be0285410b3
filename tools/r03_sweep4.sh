#!/bin/bash
O=gpurun_out/r03d; mkdir -p $O
export PYTHONPATH=$PWD
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{
EDHIP_WAVE=0 TAG="wave=0" ITERS=30 T 256 3 5
for occ in 3 4; do
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ TAG="wave=3 occ=$occ" ITERS=30 T 256 3 5
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_WAVE_STRIP=2 TAG="wave=3 occ=$occ strip=2" ITERS=30 T 256 3 5
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_TILE_DBG=11 TAG="wave=3 occ=$occ dbg=11" ITERS=20 T 256 3 5
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_TILE_DBG=2 TAG="wave=3 occ=$occ dbg=2" ITERS=20 T 256 3 5
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_TILE_DBG=1 TAG="wave=3 occ=$occ dbg=1" ITERS=20 T 256 3 5
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ TAG="wave=3 occ=$occ" ITERS=20 T 256 3 10
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_PRINT_SPILL=1 ITERS=2 timeout 120 python tools/time_k12.py 256 3 10 2>&1 | grep "edhip:" | sort | uniq -c
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_PRINT_SPILL=1 ITERS=2 timeout 120 python tools/time_k12.py 256 3 5 2>&1 | grep "edhip:" | sort | uniq -c
done
( EDHIP_WAVE=3 EDHIP_WAVE_OCC=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or ragged or hot or box or cfg3 or cfg5 or stale" 2>&1 | tail -5 )
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
