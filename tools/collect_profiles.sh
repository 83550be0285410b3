#!/bin/bash
# One pass over everything the committed profiles/ come from (run on the GPU box via gpurun).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01 -o r01 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_r01.log 2>&1
cd $R; python tools/prof_summary.py $(ls gpurun_out/prof_r01/*.db | head -1) gpurun_out/prof_r01_stats.txt > /dev/null
bash tools/pmc_k1.sh
python tools/pmc_summary.py gpurun_out/pmc0 > gpurun_out/pmc_summary.txt
BIG=1 python tools/bench_misc.py > gpurun_out/bench_misc.json 2>/dev/null
tail -1 gpurun_out/bench_r01.json
python tools/bench_matrix.py > gpurun_out/bench_matrix.txt 2>/dev/null
