#!/bin/bash
# round 5: K2 flush with incremental row offsets -- parity + timing
cd /root/repo; O=gpurun_out/r05n; mkdir -p $O
python tools/time_k12.py 256 3 5 > $O/k12.txt 2>/dev/null; python tools/time_k12.py 256 3 10 >> $O/k12.txt 2>/dev/null; python tools/time_k12.py 256 1 5 >> $O/k12.txt 2>/dev/null; python tools/time_k12.py 256 2 5 >> $O/k12.txt 2>/dev/null; python tools/time_k12.py 128 3 5 >> $O/k12.txt 2>/dev/null; cat $O/k12.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grad or golden or cfg or ragged or stale or zero" > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 400 python tests/fuzz/fuzz_hot.py 9201 300 2>&1 | grep -v amdgpu.ids | tail -2 > $O/fuzz_hot.txt; cat $O/fuzz_hot.txt
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05n/bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'])
PY
