#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03lat; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest4.txt
timeout 600 python tools/latency_small.py > $O/lat4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in rt32 cfg1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -o p -- python $GRAFT_REPO_ROOT/tools/small_case.py $c 300 > /dev/null 2>&1
  f=$(find /tmp/p_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; python $GRAFT_REPO_ROOT/tools/kernel_stats_csv.py $f 2>/dev/null | head -16
done > $O/kstats4.txt 2>&1
cat $O/pytest4.txt $O/lat4.txt; cut -c1-180 $O/kstats4.txt
