export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; rm -rf $O; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r06 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --repeats 0 > $O/prof.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof/r06_kernel_stats.csv > $O/kernel_stats.txt
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
for s in 5 10; do python tools/geo_phases.py $s 2>&1 | grep -v amdgpu; done > $O/geo_phases.txt
cp /tmp/ship.so elasticdeform_amd/libedhip.so
cat $O/geo_phases.txt; head -14 $O/kernel_stats.txt | cut -c1-150; python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06a/bench_cfg2.json'))
print(d['ms_per_step'], d.get('repeat_ms_per_step'), d['phases_ms'], d['roofline']['avg_launch_us'], d['north_star_kernel']['avg_launch_us'], d['stress']['ms_per_step'], d['fresh_grid']['sigma_5'])
PY
