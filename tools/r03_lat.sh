#!/bin/bash
export PYTHONPATH=$PWD
O=gpurun_out/r03lat; mkdir -p $O
timeout 600 python tools/latency_small.py > $O/lat.txt 2>&1
cat $O/lat.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest.txt
cat $O/pytest.txt
