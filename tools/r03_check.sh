#!/bin/bash
O=gpurun_out/r03k; mkdir -p $O
export PYTHONPATH=$PWD
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -25 ) > $O/tests.txt
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.txt
tail -6 $O/tests.txt; python -c "
import json; d=json.loads(open('$O/bench.txt').read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'], d['stress']['ms_per_step'])"
