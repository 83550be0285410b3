#!/bin/bash
# One pass over everything the committed profiles/r02_* files come from (run on the GPU box via gpurun):
#   COMMIT=$(git rev-parse --short HEAD) ; gpurun -- "COMMIT=$COMMIT bash tools/collect_profiles_r02.sh"
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
# kernel trace of the bench command
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r02 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
cd $R; python tools/prof_summary.py $(ls $O/prof/*.db | head -1) $O/kernel_stats.txt > /dev/null
# counters: SQ groups, then FETCH_SIZE and WRITE_SIZE in passes of their own
OUTNAME=r02/pmc ABLV=0 bash tools/pmc_hot.sh
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ITERS=6 rocprofv3 --kernel-trace --pmc $c -d $O/pmc/$c -o p --output-format csv -- python $R/tools/time_k12.py > $O/pmc/$c.log 2>&1
done
cd $R; python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/hbm_traffic.py $O/pmc "${COMMIT:-unknown}" > $O/hbm_traffic.json
# ablations of the two hot kernels
ABL="0 2 4 6 32 46" bash tools/ablate_hot.sh > $O/ablate_k1.txt 2>&1
for v in 0 128 64 192; do EDHIP_TILE_DBG=$v TAG="dbg=$v" python tools/time_k12.py 2>/dev/null; done > $O/ablate_k2.txt
# other configurations
{ python tools/time_k12.py 256 3 10; python tools/time_k12.py 256 1 5; python tools/time_k12.py 256 2 5; python tools/time_k12.py 256 4 5; python tools/time_k12.py 256 5 5; python tools/time_k12.py 128 3 5;
  TAG=one python tools/time_batch.py 32; EDHIP_BATCH_LOOP=1 TAG=loop python tools/time_batch.py 32; TAG=one python tools/time_batch.py 64;
  python tools/time_small.py; } > $O/misc.txt 2>/dev/null
tools/ubench_issue.bin valu > $O/ubench_valu.txt 2>&1
timeout 120 tools/ubench_issue.bin lds > $O/ubench_lds.txt 2>&1
tail -c 400 $O/bench_cfg2.json
