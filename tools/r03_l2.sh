#!/bin/bash
export PYTHONPATH=$PWD
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
for s in 10 5; do
TAG="L2 48K x512 (default)" ITERS=20 T 256 3 $s
for cfg in "32 768" "28 1024" "24 1024" "24 1280" "20 1280" "48 1024"; do set -- $cfg
  EDHIP_L2_BOX_KB=$1 EDHIP_L2_WGS=$2 TAG="L2 ${1}K x$2" ITERS=20 T 256 3 $s
done; done
EDHIP_L2_BOX_KB=24 EDHIP_L2_WGS=1024 EDHIP_PRINT_SPILL=1 ITERS=2 timeout 120 python tools/time_k12.py 256 3 10 2>&1 | grep "edhip:" | sort | uniq -c
EDHIP_L2_BOX_KB=24 EDHIP_L2_WGS=1024 TAG="L2 24K x1024 order4" ITERS=20 T 256 4 10
TAG="L2 default order4" ITERS=20 T 256 4 10
