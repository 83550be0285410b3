#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "crop or cfg4 or baseline_configs or source_box" 2>&1 | tail -25 ) > $O/pytest.txt
( timeout 300 python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err )
( timeout 300 python bench.py --workload cfg4 > $O/bench_cfg4b.json 2> $O/bench_cfg4b.err )
cd /tmp
for w in auto off; do
  WINDOW=$w WHICH=fwd rocprofv3 --kernel-trace --stats -d $O/$w -o p --output-format csv -- python $R/tools/cfg4_calls.py 10 > $O/$w.log 2>&1
  python $R/tools/kernel_stats_csv.py $O/$w/p_kernel_stats.csv > $O/stats_$w.txt 2>/dev/null
done
cd $R
cat $O/pytest.txt; for f in bench_cfg4 bench_cfg4b; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'], d['value'], d['crop_window'])"; done
for w in auto off; do echo "== window $w (forward x11)"; head -12 $O/stats_$w.txt | cut -c1-180; done
