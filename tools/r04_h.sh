#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
for w in on off; do
  WINDOW=$w WHICH=fwd rocprofv3 --kernel-trace --stats -d $O/$w -o p --output-format csv -- python $R/tools/cfg4_calls.py 10 > $O/$w.log 2>&1
  python $R/tools/kernel_stats_csv.py $O/$w/p_kernel_stats.csv > $O/stats_$w.txt 2>/dev/null || cp $O/$w/p_kernel_stats.csv $O/stats_$w.txt
done
for w in on off; do echo "== window $w (forward x11)"; head -14 $O/stats_$w.txt | cut -c1-200; done
