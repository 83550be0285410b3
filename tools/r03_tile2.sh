#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03tile; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prefilter or filter or golden or ragged or int" 2>&1 | tail -4 ) > $O/pytest.txt
( timeout 400 python tests/fuzz/fuzz_filter.py 61 500 2>&1 | tail -3 ) >> $O/pytest.txt
( timeout 400 python tests/fuzz/fuzz_filter.py 62 500 2>&1 | tail -3 ) >> $O/pytest.txt
timeout 800 python tools/time_matrix.py 2>&1 | grep -v amdgpu.ids > $O/time_matrix.txt
cat $O/pytest.txt; grep -E "2d 1024|order 5" $O/time_matrix.txt
