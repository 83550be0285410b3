# dev tool: rocprofv3 kernel stats of the benchmark step at sigma 10 and 15 (tools/prof_step.py), shipped library
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06sig; rm -rf $O; mkdir -p $O
for s in 10 15; do
  cd /tmp && rocprofv3 --kernel-trace --stats -d $O/s$s -o p --output-format csv -- python $R/tools/prof_step.py $s 20 > $O/s$s.log 2>&1
  cd $R; echo "== sigma $s"; python tools/kernel_stats_csv.py $O/s$s/p_kernel_stats.csv | cut -c1-160 | head -16
done
