#!/bin/bash
O=gpurun_out/r03e; mkdir -p $O
export PYTHONPATH=$PWD
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{
EDHIP_WAVE=0 TAG="wave=0" ITERS=30 T 256 3 5
for occ in 3 4; do
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ TAG="wave=3 occ=$occ" ITERS=30 T 256 3 5
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_TILE_DBG=16 TAG="wave=3 occ=$occ dbg=16(zy)" ITERS=30 T 256 3 5
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ BOXES=0 TAG="wave=3 occ=$occ noboxes" ITERS=30 T 256 3 5
done
( EDHIP_WAVE=3 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or ragged or hot or box or cfg3 or cfg5 or stale" 2>&1 | tail -3 )
EDHIP_WAVE=3 OUTNAME=r03e/pmc_wave bash tools/pmc_hot.sh
grep -A9 "wave_grad" gpurun_out/r03e/pmc_wave/summary.txt
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
