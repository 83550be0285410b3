#!/bin/bash
O=gpurun_out/r03c; mkdir -p $O
export PYTHONPATH=$PWD
{
EDHIP_WAVE=3 timeout 120 python tools/wg_timeline.py
EDHIP_WAVE=3 EDHIP_TILE_DBG=11 timeout 120 python tools/wg_timeline.py
EDHIP_WAVE=3 timeout 120 python tools/wg_timeline.py grad
rocm-smi --showclocks 2>/dev/null | head -20
} > $O/sweep2.txt 2>&1
cat $O/sweep2.txt
