#!/bin/bash
# last step of round 5: PMC passes + traffic stamp on the final sources, the bench line, the whole GPU suite
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05final; rm -rf $O; mkdir -p $O; export PYTHONPATH=$R; cd $R
OUTNAME=r05final/pmc bash tools/pmc_k1.sh
cd $R; cp $O/pmc/summary.txt $O/pmc_summary.txt
python tools/hbm_traffic.py $O/pmc "${COMMIT:-unknown}" > $O/hbm_traffic.json
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
python bench.py --steps 20 --warmup 5 2> $O/bench.err | grep '^{' > $O/bench.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r05 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/prof.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof/r05_kernel_stats.csv > $O/kernel_stats.txt
timeout 300 python tools/latency_small.py 2>&1 | grep -v amdgpu.ids > $O/host_latency.txt
timeout 600 python tools/time_matrix.py 2>&1 | grep -v amdgpu.ids > $O/time_matrix.txt
timeout 300 python tools/time_big_grid.py 2>&1 | grep -v amdgpu.ids > $O/big_grid.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
{ timeout 400 python tests/fuzz/fuzz_api.py 9706 300; timeout 400 python tests/fuzz/fuzz_api.py 9806 300; timeout 400 python tests/fuzz/fuzz_int.py 9807 200; } 2>&1 | grep -v amdgpu.ids | grep -A2 "cases\|FAIL" > $O/fuzz_api.txt; cat $O/fuzz_api.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05final/bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d['north_star_kernel']['avg_launch_us'], d['north_star_kernel']['traffic'])
PY
