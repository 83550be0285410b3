# usage: tools/prof_filter.sh <tag> [env...]   -- rocprofv3 kernel durations of tools/bench_filter.py
export TMPDIR=/tmp; R=$PWD; tag=$1; shift
cd /tmp && env "$@" rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_$tag -o t -- python $R/tools/bench_filter.py > $R/gpurun_out/pf_$tag.log 2>&1
cd $R; python tools/prof_summary.py $(ls gpurun_out/pf_$tag/*.db | head -1) gpurun_out/pf_${tag}_stats.txt | cut -c1-50,105-200 | head -9
