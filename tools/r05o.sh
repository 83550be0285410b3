#!/bin/bash
# round 5: grid prefilter inside the tables launch -- whole GPU suite, bench, small-volume latency
cd /root/repo; O=gpurun_out/r05o; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05o/bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'], d.get('stress'), d.get('fresh_grid'))
PY
timeout 300 python tools/latency_small.py > $O/host_latency.txt 2>/dev/null; cat $O/host_latency.txt
timeout 300 python tests/fuzz/fuzz_round4.py 9301 150 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tests/fuzz/fuzz_api.py 9302 200 2>&1 | grep -v amdgpu.ids | tail -2
