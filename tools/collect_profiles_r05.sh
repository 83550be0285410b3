#!/bin/bash
# Final collection of round 5: COMMIT=$(git rev-parse --short HEAD) gpurun -- "COMMIT=$COMMIT bash tools/collect_profiles_r05.sh"
# shipped library: PMC passes + HBM traffic / busy fractions (stamped), bench lines (cfg2 with fresh_grid / stress, cfg4,
# cfg5, bf16, 2 ranks on the one GPU, collective leg), rocprofv3 kernel stats of the same bench command, time matrices;
# profiling build: K1 against the round-4 kernel on this box, K1's ablation / sensitivity table, per-wave phase clocks,
# occupancy sweep, K2's lane-map experiment; counters build: K1's tile classes and redone windows.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
cd $R
OUTNAME=r05/pmc bash tools/pmc_k1.sh
cd $R; cp $O/pmc/summary.txt $O/pmc_summary.txt
python tools/hbm_traffic.py $O/pmc "${COMMIT:-unknown}" > $O/hbm_traffic.json
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
EDHIP_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2> $O/bench_2r.err | grep '^{' > $O/bench_cfg5_2ranks_gloo.json
EDHIP_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 --collective 2> $O/bench_coll.err | grep '^{' > $O/bench_cfg5_collective_gloo.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r05 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/prof.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof/r05_kernel_stats.csv > $O/kernel_stats.txt
cd /tmp
WINDOW=auto rocprofv3 --kernel-trace --stats -d $O/cfg4_auto -o p --output-format csv -- python $R/tools/cfg4_calls.py 10 > $O/cfg4_auto.log 2>&1
python $R/tools/kernel_stats_csv.py $O/cfg4_auto/p_kernel_stats.csv > $O/cfg4_stats_auto.txt 2>/dev/null
cd $R
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{ for o in 1 2 3; do T 256 $o 5; done; T 256 3 10; T 256 3 15; T 128 3 5; T 64 3 5; } > $O/misc.txt 2>/dev/null
timeout 300 python tools/time_4d.py 2>&1 | grep -v amdgpu.ids | grep grad >> $O/misc.txt
timeout 300 python tools/time_big_grid.py 2>&1 | grep -v amdgpu.ids >> $O/misc.txt
timeout 600 python tools/time_matrix.py 2>&1 | grep -v amdgpu.ids > $O/time_matrix.txt
timeout 300 python tools/latency_small.py 2>&1 | grep -v amdgpu.ids > $O/host_latency.txt
# ---- fuzz campaign on this build
{ timeout 600 python tests/fuzz/fuzz_hot.py 9001 400; timeout 600 python tests/fuzz/fuzz_parity.py 9002 250; timeout 300 python tests/fuzz/fuzz_int.py 9003 200;
  timeout 600 python tests/fuzz/fuzz_round4.py 9004 200; timeout 300 python tests/fuzz/fuzz_filter.py 9005 400; timeout 300 python tests/fuzz/fuzz_api.py 9006 200; } 2>&1 | grep -v amdgpu.ids | grep "cases\|FAIL" > $O/fuzz.txt
# ---- profiling build
cp elasticdeform_amd/libedhip.so /tmp/libedhip_ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
echo "# K1 of round 5 (deform_k1.hip) against the kernel it replaced (experiments/deform_hot_r4.hip, EDHIP_K1_OLD=1), one box, tools/time_k12.py (profiling build)"
for a in "256 3 5" "256 3 10" "256 3 15" "256 1 5" "256 2 5" "128 3 5"; do
  TAG="K1 round 5" ITERS=30 T $a; TAG="K1 round 4" EDHIP_K1_OLD=1 ITERS=30 T $a
done
} > $O/k1_vs_r4.txt 2>&1
{
echo "# K1 (k1_fwd_kernel<3, false>, profiling build): level-1 launch with parts of the kernel switched off or executed twice (EDHIP_TILE_DBG bits)"
TAG="full                                   " ITERS=30 T 256 3 5
TAG="no gather (1<<17)                      " EDHIP_TILE_DBG=131072 ITERS=30 T 256 3 5
TAG="no staging (1<<18)                     " EDHIP_TILE_DBG=262144 ITERS=30 T 256 3 5
TAG="no stores (1<<20)                      " EDHIP_TILE_DBG=1048576 ITERS=30 T 256 3 5
TAG="no gather, no staging                  " EDHIP_TILE_DBG=393216 ITERS=30 T 256 3 5
TAG="no gather, no staging, no stores       " EDHIP_TILE_DBG=1441792 ITERS=30 T 256 3 5
TAG="displacement twice (1<<19)             " EDHIP_TILE_DBG=524288 ITERS=30 T 256 3 5
TAG="weights + gather twice (1<<23)         " EDHIP_TILE_DBG=8388608 ITERS=30 T 256 3 5
TAG="every tile on general coordinates (1<<16)" EDHIP_TILE_DBG=65536 ITERS=30 T 256 3 5
TAG="gather reads merged by the backend (split 0)" EDHIP_K1_SPLIT=0 ITERS=30 T 256 3 5
TAG="3 workgroups per CU (52 KB)            " EDHIP_HOT_FWD_KB=52 ITERS=30 T 256 3 5
TAG="2 workgroups per CU (64 KB)            " EDHIP_HOT_FWD_KB=64 ITERS=30 T 256 3 5
TAG="strips of 2 tiles                      " EDHIP_STRIP=2 ITERS=30 T 256 3 5
} > $O/ablate_k1.txt 2>&1
{ python tools/k1_phases.py 5 3; python tools/k1_phases.py 10 3; python tools/k1_phases.py 5 1; } 2>&1 | grep -v amdgpu.ids > $O/k1_phases.txt
{
echo "# K2 (hot_grad_kernel): lane -> voxel maps and cell layouts suggested by tools/sim/conflicts_k2_4wave.py, measured (profiling build)"
TAG="shipped: x two apart, rows y / y + 2, pitch 8 odd" ITERS=30 T 256 3 5
TAG="16 consecutive x per 16 lanes (1<<21)           " EDHIP_TILE_DBG=2097152 ITERS=30 T 256 3 5
TAG="pitch 32 (1<<22)                                " EDHIP_TILE_DBG=4194304 ITERS=30 T 256 3 5
TAG="16 consecutive x + pitch 32                     " EDHIP_TILE_DBG=6291456 ITERS=30 T 256 3 5
TAG="shipped, no flush (64)                          " EDHIP_TILE_DBG=64 ITERS=30 T 256 3 5
TAG="shipped, no scatter (128)                       " EDHIP_TILE_DBG=128 ITERS=30 T 256 3 5
} > $O/k2_lane_maps.txt 2>&1
cp tools/libedhip_stats.so elasticdeform_amd/libedhip.so
{
echo "# K1 tile classes and redone windows (counters build, tools/k1_stats.py: side order sigma [control points] [mode])"
for a in "256 3 5" "256 3 10" "256 3 15" "256 3 5 8" "256 3 5 3" "256 1 5" "256 2 5" "256 3 5 5 constant" "256 3 5 5 nearest" "256 3 5 5 reflect" "256 3 5 5 wrap" "128 3 5" "200 3 5" "64 3 3 13"; do
  timeout 100 python tools/k1_stats.py $a 2>&1 | tail -1
done
} > $O/k1_stats.txt 2>&1
cp /tmp/libedhip_ship.so elasticdeform_amd/libedhip.so
tail -c 1200 $O/bench_cfg2.json; echo; cat $O/fuzz.txt; cat $O/k1_vs_r4.txt
