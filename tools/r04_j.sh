#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -12 ) > $O/pytest.txt
( timeout 300 python bench.py > $O/bench.json 2> $O/bench.err )
( timeout 300 python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err )
cd /tmp
WINDOW=auto WHICH=fwd rocprofv3 --kernel-trace --stats -d $O/auto -o p --output-format csv -- python $R/tools/cfg4_calls.py 10 > $O/auto.log 2>&1
python $R/tools/kernel_stats_csv.py $O/auto/p_kernel_stats.csv > $O/stats_auto.txt 2>/dev/null
cd $R
cat $O/pytest.txt; python -c "
import json; d=json.load(open('$O/bench.json')); print('cfg2', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'], d['stress']['ms_per_step'], d['fresh_grid']['sigma_5']['ms_per_step'], d['phases_ms'])
d=json.loads(open('$O/bench_cfg4.json').read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'], d['value'], d['crop_window'])"
head -14 $O/stats_auto.txt | cut -c1-170
