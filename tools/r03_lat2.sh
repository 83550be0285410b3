#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r03lat; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in rt32 rt128 cfg1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -o p -- python $GRAFT_REPO_ROOT/tools/small_case.py $c 300 > /dev/null 2>&1
  f=$(find /tmp/p_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; python $GRAFT_REPO_ROOT/tools/kernel_stats_csv.py $f 2>/dev/null | head -25 || head -20 $f
done > $O/kstats.txt 2>&1
cat $O/kstats.txt | cut -c1-220
