#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; rm -rf $O; mkdir -p $O; cd /tmp; export PYTHONPATH=$R
cp $R/tools/libedhip_exp.so $R/elasticdeform_amd/libedhip.so
for dbg in 0 1 2 3; do
EDHIP_4D_DBG=$dbg rocprofv3 --kernel-trace --stats -d $O/p$dbg -o p --output-format csv -- python $R/tools/prof_4d.py 0 > $O/p$dbg.log 2>&1
echo "dbg $dbg"; python $R/tools/kernel_stats_csv.py $O/p$dbg/p_kernel_stats.csv | grep fast4
done
