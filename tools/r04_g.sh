#!/bin/bash
# new golden test (cfg2 gradient at 256^3), bench modes: default line (fresh_grid), cfg4, 2 ranks on one GPU (gloo), collective leg
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
export PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "baseline_configs" 2>&1 | tail -6 ) > $O/pytest.txt
( timeout 300 python bench.py > $O/bench.json 2> $O/bench.err )
( timeout 300 python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err )
( EDHIP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 > $O/bench_2rank.json 2> $O/bench_2rank.err )
( EDHIP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 --collective > $O/bench_coll.json 2> $O/bench_coll.err )
cat $O/pytest.txt
python - <<PY
import json
d=json.load(open('$O/bench.json')); print('cfg2', d['ms_per_step'], d['value'], 'fresh', d.get('fresh_grid'), 'stress', d['stress']['ms_per_step'])
for f in ('bench_cfg4','bench_2rank','bench_coll'):
    try:
        t=open('$O/'+f+'.json').read().strip().splitlines()
        d=json.loads(t[-1]); print(f, d['n_gpus'], d['ms_per_step'], d['value'], d.get('crop_window'), d.get('link_GBps'))
    except Exception as e:
        print(f, 'FAILED', e); print(open('$O/'+f+'.err').read()[-1500:])
PY
