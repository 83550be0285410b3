#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03rel; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "channel_last or strided or fortran or golden or ragged or lane" 2>&1 | tail -6 ) > $O/pytest.txt
timeout 600 python tools/time_matrix.py 2>&1 | grep -E "ch-last|3d\+ch" > $O/tm.txt
cat $O/pytest.txt $O/tm.txt
