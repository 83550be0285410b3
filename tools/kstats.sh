#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel statistics of a python script of this repo, printed as the table kept under profiles/
#   tools/kstats.sh <tag> <script relative to the repo> [args...]
tag=$1; shift
script=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python $R/$script "$@" > $out/cmd.log 2>&1)
f=$(find $out -name "*kernel_stats.csv" | head -1)
python $R/tools/kernel_stats_csv.py $f | cut -c1-200 | head -${KSTATS_ROWS:-14}
