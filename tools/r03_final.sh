#!/bin/bash
# final validation of a build on the GPU box: full GPU suite, smoke, fuzzers, bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03final; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) >> $O/pytest.txt
for s in 101 102; do ( timeout 400 python tests/fuzz/fuzz_hot.py $s 150 2>&1 | tail -3 ) > $O/fuzz_hot_$s.txt; done
( timeout 500 python tests/fuzz/fuzz_int.py 103 150 2>&1 | tail -3 ) > $O/fuzz_int.txt
( timeout 300 python tests/fuzz/fuzz_filter.py 104 300 2>&1 | tail -3 ) > $O/fuzz_filter.txt
( timeout 300 python tests/fuzz/fuzz_parity.py 105 150 2>&1 | tail -3 ) > $O/fuzz_parity.txt
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/pytest.txt; tail -n 1 $O/fuzz_*.txt; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'], d['stress']['ms_per_step'], d['commit'])"
