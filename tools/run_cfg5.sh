export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p5 -o p --output-format csv -- python $R/bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --repeats 0 > /tmp/p5.log 2>&1
cd $R; python tools/kernel_stats_csv.py /tmp/p5/p_kernel_stats.csv | cut -c1-150 | head -14
