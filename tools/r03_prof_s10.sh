#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
export PYTHONPATH=$R
ITERS=10 rocprofv3 --kernel-trace --stats -d $O/s10 -o p --output-format csv -- python $R/tools/time_k12.py 256 3 10 > $O/s10.log 2>&1
python $R/tools/prof_summary.py $O/s10 > $O/s10_stats.txt 2>&1 || ls -R $O/s10 | head
head -30 $O/s10_stats.txt
