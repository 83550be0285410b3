#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03i32; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "integer or label or golden" 2>&1 | tail -6 ) > $O/pytest.txt
( timeout 600 python tests/fuzz/fuzz_int.py 71 200 2>&1 | tail -8 ) > $O/fuzz71.txt
( timeout 600 python tests/fuzz/fuzz_int.py 72 200 2>&1 | tail -8 ) > $O/fuzz72.txt
timeout 300 python tools/time_int.py 256 2>&1 | tail -5 > $O/time.txt
cat $O/pytest.txt $O/fuzz71.txt $O/fuzz72.txt $O/time.txt
