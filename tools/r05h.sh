#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; rm -rf $O; mkdir -p $O; cd /tmp; export PYTHONPATH=$R
for pf in 0 1; do
rocprofv3 --kernel-trace --stats -d $O/p$pf -o p --output-format csv -- python $R/tools/prof_4d.py $pf > $O/p$pf.log 2>&1
python $R/tools/kernel_stats_csv.py $O/p$pf/p_kernel_stats.csv | head -8
done
