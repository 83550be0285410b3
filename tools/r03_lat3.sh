#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03lat; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prefilter or filter or golden or ragged or int" 2>&1 | tail -5 ) > $O/pytest3.txt
( timeout 300 python tests/fuzz/fuzz_filter.py 51 300 2>&1 | tail -3 ) >> $O/pytest3.txt
timeout 600 python tools/latency_small.py > $O/lat3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in rt32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -o p -- python $GRAFT_REPO_ROOT/tools/small_case.py $c 300 > /dev/null 2>&1
  f=$(find /tmp/p_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; python $GRAFT_REPO_ROOT/tools/kernel_stats_csv.py $f 2>/dev/null | head -25
done > $O/kstats3.txt 2>&1
cat $O/pytest3.txt $O/lat3.txt; cut -c1-200 $O/kstats3.txt
