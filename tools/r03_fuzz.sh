#!/bin/bash
O=gpurun_out/r03fuzz; mkdir -p $O
export PYTHONPATH=$PWD
( timeout 500 python tests/fuzz/fuzz_int.py 1 150 2>&1 | tail -25 ) > $O/int1.txt
( timeout 500 python tests/fuzz/fuzz_int.py 2 150 2>&1 | tail -25 ) > $O/int2.txt
( timeout 400 python tests/fuzz/fuzz_hot.py 31 120 2>&1 | tail -15 ) > $O/hot31.txt
( timeout 400 python tests/fuzz/fuzz_hot.py 32 120 2>&1 | tail -15 ) > $O/hot32.txt
( timeout 300 python tests/fuzz/fuzz_filter.py 33 300 2>&1 | tail -10 ) > $O/filter33.txt
( timeout 300 python tests/fuzz/fuzz_parity.py 34 150 2>&1 | tail -10 ) > $O/parity34.txt
tail -3 $O/*.txt
