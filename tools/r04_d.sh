#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg2 or cfg3 or cfg5 or gradient or grad" 2>&1 | tail -4 ) > $O/pytest.txt
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{ timeout 200 python tools/g2_phases.py 5;
EDHIP_RECORDS=1 TAG="new    " ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="old    " ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
EDHIP_RECORDS=1 TAG="new s10" ITERS=20 timeout 200 python tools/time_k12.py 256 3 10
} 2>&1 | grep -v amdgpu.ids > $O/phases.txt
cat $O/pytest.txt $O/phases.txt
