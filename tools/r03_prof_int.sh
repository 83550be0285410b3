#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d $O/int -o p --output-format csv -- python $R/tools/time_int.py > $O/int.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/int/p_kernel_stats.csv')))
for r in rows[:12]:
    print("%-8s calls %-4s avg %9.1f us  min %9.1f max %9.1f  %s" % (r['Percentage'], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Name'][:100]))
PY
