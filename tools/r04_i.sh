#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "crop or cfg4 or baseline_configs or source_box" 2>&1 | tail -25 ) > $O/pytest.txt
( timeout 300 python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err )
( timeout 300 python tools/time_crop_window.py > $O/time_crop.txt 2>&1 )
cat $O/pytest.txt; python -c "
import json; d=json.loads(open('$O/bench_cfg4.json').read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'], d['value'], d['crop_window'])"; tail -5 $O/bench_cfg4.err; tail -12 $O/time_crop.txt
