#!/bin/bash
# dev tool: ablation timings of the hot forward kernel (EDHIP_HOT_ABL, see deform_hot.hip)
for a in ${ABL:-0 2 4 6 12 14 46 110 64 32}; do
  EDHIP_HOT_ABL=$a TAG="ABL=$a" python tools/time_k12.py 256 3 5 2>/dev/null
done
