#!/bin/bash
# Final collection of round 3 (shipped library only): COMMIT=$(git rev-parse --short HEAD) gpurun -- "COMMIT=$COMMIT bash tools/collect_profiles_r03b.sh"
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
export PYTHONPATH=$R
cd $R
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r03 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/prof.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof/r03_kernel_stats.csv > $O/kernel_stats.txt
OUTNAME=r03b/pmc bash tools/pmc_hot.sh
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ITERS=6 rocprofv3 --kernel-trace --pmc $c -d $O/pmc/$c -o p --output-format csv -- python $R/tools/time_k12.py > $O/pmc/$c.log 2>&1
done
cd $R; python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/hbm_traffic.py $O/pmc "${COMMIT:-unknown}" > $O/hbm_traffic.json
{ for o in 1 2 3 4 5; do T 256 $o 5; done; for o in 3 4 5; do T 256 $o 10; done; T 256 3 15; T 128 3 5
  TAG=one python tools/time_batch.py 32; TAG=one python tools/time_batch.py 64
  python tools/time_int.py; } > $O/misc.txt 2>/dev/null
timeout 600 python tools/time_matrix.py 2>&1 | grep -v amdgpu.ids > $O/time_matrix.txt
timeout 600 python tools/latency_small.py 2>&1 | grep -v amdgpu.ids > $O/latency.txt
tail -c 700 $O/bench_cfg2.json
