#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; rm -rf $O; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_deformed" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python tools/time_4d.py 2>&1 | grep -v amdgpu | grep "grad" > $O/time_4d.txt; cat $O/time_4d.txt
cd /tmp
for a in "0 f32" "0 f64"; do
rocprofv3 --kernel-trace --stats -d $O/p -o p --output-format csv -- python $R/tools/prof_4d.py $a > $O/p.log 2>&1
python $R/tools/kernel_stats_csv.py $O/p/p_kernel_stats.csv | grep fast4
done
