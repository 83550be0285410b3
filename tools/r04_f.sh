#!/bin/bash
# self-serve: level 1 takes its own oversize tiles, no spill launches
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -8 ) > $O/pytest.txt
( timeout 300 python bench.py > $O/bench.json 2> $O/bench.err )
cp elasticdeform_amd/libedhip.so /tmp/libedhip_ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
TAG="auto       " ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="selfserve=0" EDHIP_SELF_SERVE=0 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="selfserve=1" EDHIP_SELF_SERVE=1 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="s10 auto   " ITERS=20 timeout 200 python tools/time_k12.py 256 3 10
TAG="s10 ss=1   " EDHIP_SELF_SERVE=1 ITERS=10 timeout 300 python tools/time_k12.py 256 3 10
# correctness of the self-served tiles where there are many of them: sigma 10, forced
EDHIP_SELF_SERVE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg2 or cfg3 or spill or level" 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids > $O/time.txt
cp /tmp/libedhip_ship.so elasticdeform_amd/libedhip.so
cat $O/pytest.txt $O/time.txt; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'], d['stress']['ms_per_step'], d['phases_ms'])"
