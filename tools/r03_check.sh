#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
export PYTHONPATH=$PWD
( timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -25 ) > $O/tests.txt
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 ) > $O/bench.txt
cat $O/tests.txt | tail -8; cat $O/bench.txt
