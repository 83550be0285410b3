#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest_final.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> $O/pytest_final.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof2 -o r03 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/prof2.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof2/r03_kernel_stats.csv > $O/kernel_stats.txt
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
cat $O/pytest_final.txt; head -8 $O/kernel_stats.txt | cut -c1-150; tail -c 400 $O/prof2.log
