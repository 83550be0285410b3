#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03hint; mkdir -p $O
cp elasticdeform_amd/libedhip.so /tmp/ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
timeout 800 python tools/cmp_levels.py 2>&1 | grep -v amdgpu > $O/cmp.txt
cp /tmp/ship.so elasticdeform_amd/libedhip.so
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
( timeout 400 python tests/fuzz/fuzz_hot.py 81 150 2>&1 | tail -6 ) >> $O/pytest.txt
( timeout 400 python tests/fuzz/fuzz_hot.py 82 150 2>&1 | tail -6 ) >> $O/pytest.txt
( timeout 400 python tests/fuzz/fuzz_parity.py 83 150 2>&1 | tail -3 ) >> $O/pytest.txt
grep -v " 0 of" $O/cmp.txt; echo "---"; cat $O/pytest.txt
