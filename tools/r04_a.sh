#!/bin/bash
# round 4, first GPU call: the records route (K1 writes coordinate records, hot_grad2_kernel reads them)
# -- GPU suite on the shipped build, bench line, then old route against new route in the profiling build
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests -m gpu -q --maxfail=40 2>&1 | tail -60 ) > $O/pytest.txt
( timeout 300 python bench.py > $O/bench.json 2> $O/bench.err )
cp elasticdeform_amd/libedhip.so /tmp/libedhip_ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
for s in 5 10; do
  EDHIP_RECORDS=1 TAG="new(rec)      " timeout 200 python tools/time_k12.py 256 3 $s
  TAG="old(boxes)    " timeout 200 python tools/time_k12.py 256 3 $s
done
EDHIP_RECORDS=1 TAG="new standalone" BOXES=0 timeout 200 python tools/time_k12.py 256 3 5
TAG="old standalone" BOXES=0 timeout 200 python tools/time_k12.py 256 3 5
for o in 1 2; do
  EDHIP_RECORDS=1 TAG="new(rec) " timeout 200 python tools/time_k12.py 256 $o 5
  TAG="old      " timeout 200 python tools/time_k12.py 256 $o 5
done
EDHIP_RECORDS=1 TAG="new 128 " timeout 200 python tools/time_k12.py 128 3 5
TAG="old 128 " timeout 200 python tools/time_k12.py 128 3 5
} > $O/time_k12.txt 2>&1
cp /tmp/libedhip_ship.so elasticdeform_amd/libedhip.so
cat $O/pytest.txt | tail -40; cat $O/time_k12.txt; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'], d['stress']['ms_per_step'], d['phases_ms'])"
tail -5 $O/bench.err
