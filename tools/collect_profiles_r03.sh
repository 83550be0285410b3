#!/bin/bash
# One pass over everything the committed profiles/r03_* files come from (run on the GPU box via gpurun):
#   COMMIT=$(git rev-parse --short HEAD); gpurun -- "COMMIT=$COMMIT bash tools/collect_profiles_r03.sh"
# Part A uses the shipped library (elasticdeform_amd/libedhip.so); part B swaps in the profiling build
# (tools/libedhip_exp.so = `make EXPERIMENTS=1`, built in the container) ON THE BOX'S COPY of the tree.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
export PYTHONPATH=$R
cd $R
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
# ---------------- part A: shipped library --------------------------------------------------------------
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r03 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof/r03_kernel_stats.csv > $O/kernel_stats.txt
OUTNAME=r03/pmc bash tools/pmc_hot.sh
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ITERS=6 rocprofv3 --kernel-trace --pmc $c -d $O/pmc/$c -o p --output-format csv -- python $R/tools/time_k12.py > $O/pmc/$c.log 2>&1
done
cd $R; python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/hbm_traffic.py $O/pmc "${COMMIT:-unknown}" > $O/hbm_traffic.json
{ for o in 1 2 3 4 5; do T 256 $o 5; done; for o in 3 4 5; do T 256 $o 10; done; T 128 3 5
  TAG=one python tools/time_batch.py 32; TAG=one python tools/time_batch.py 64
  python tools/time_small.py; python tools/time_int.py; } > $O/misc.txt 2>/dev/null
# ---------------- part B: profiling build -----------------------------------------------------------------
if [ -f tools/libedhip_exp.so ]; then
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
echo "# level-1 launch, 256^3 float32 order 3 sigma 5: 4-wave kernels (wave=0) vs one wavefront per tile (wave=3)"
EDHIP_WAVE=0 TAG="wave=0" ITERS=30 T 256 3 5
for occ in 3 4; do
  EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ TAG="wave=3 occ=$occ" ITERS=30 T 256 3 5
  for d in 1 2 3 8 11; do EDHIP_WAVE=3 EDHIP_WAVE_OCC=$occ EDHIP_TILE_DBG=$d TAG="wave=3 occ=$occ dbg=$d" ITERS=20 T 256 3 5; done
done
for st in 1 2 8; do EDHIP_WAVE=3 EDHIP_WAVE_STRIP=$st TAG="wave=3 strip=$st" ITERS=20 T 256 3 5; done
for l in 10240 16384 20480; do EDHIP_WAVE=3 EDHIP_WAVE_LDS=$l TAG="wave=3 lds=$l" ITERS=20 T 256 3 5; done
echo "# other orders / sigma 10: level 1 on the 4-wave kernels (wave=0) and on the wave kernels (wave=3)"
for s in 5 10; do for o in 1 2 3 4 5; do for w in 0 3; do EDHIP_WAVE=$w TAG="wave=$w" ITERS=15 T 256 $o $s; done; done; done
} > $O/wave_vs_4wave.txt 2>&1
{ EDHIP_WAVE=3 timeout 120 python tools/wg_timeline.py; EDHIP_WAVE=3 EDHIP_TILE_DBG=11 timeout 120 python tools/wg_timeline.py; } > $O/wave_timeline.txt 2>&1
EDHIP_WAVE=3 OUTNAME=r03/pmc_wave bash tools/pmc_hot.sh
fi
tools/ubench_clock.bin > $O/ubench_clock.txt 2>&1
tools/ubench_ilp.bin > $O/ubench_ilp.txt 2>&1
tail -c 600 $O/bench_cfg2.json
