#!/bin/bash
# round 5: float64 orders 4 / 5 on the tile prefilter -- tests, fuzz, timings
cd /root/repo; O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefilter or filter or golden or ragged" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python tests/fuzz/fuzz_filter.py 9105 400 2>&1 | grep -v amdgpu.ids | tail -3 > $O/fuzz_filter.txt; cat $O/fuzz_filter.txt
timeout 300 python tools/time_matrix.py 2>/dev/null | grep -E "float64|float32  order 5" > $O/time_matrix_f64.txt; cat $O/time_matrix_f64.txt
