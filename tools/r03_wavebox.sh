#!/bin/bash
export PYTHONPATH=$PWD
cp elasticdeform_amd/libedhip.so /tmp/ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
for o in 4 5; do for s in 5 10; do
TAG="default o$o s$s" ITERS=12 T 256 $o $s
for l in 16384 20480 26624; do EDHIP_WAVE_LDS=$l TAG="lds=$l o$o s$s" ITERS=12 T 256 $o $s; done
done; done
cp /tmp/ship.so elasticdeform_amd/libedhip.so
