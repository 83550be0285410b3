#!/bin/bash
# round 5: where the time of float64 order-5 volumes goes (128^3, 256^3)
R=/root/repo; O=$R/gpurun_out/r05l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for n in 128 256; do
rocprofv3 --kernel-trace --stats -d $O/p$n -o p --output-format csv -- python $R/tools/prof_case.py "$n,$n,$n" - float64 5 10 > $O/p$n.log 2>&1
python $R/tools/kernel_stats_csv.py $O/p$n/p_kernel_stats.csv | cut -c1-220 | head -14 > $O/stats_$n.txt; cat $O/stats_$n.txt
done
