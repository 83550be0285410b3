#!/bin/bash
# Final collection of round 6: COMMIT=$(git rev-parse --short HEAD) gpurun -- "COMMIT=$COMMIT bash tools/collect_profiles_r06.sh"
# shipped library: PMC passes + fabric traffic / busy fractions (stamped), bench lines (cfg2 with repeats / fresh_grid / stress, cfg4, cfg5,
# bf16, 2 ranks on the one GPU, collective leg), rocprofv3 kernel stats of the same bench command, time matrices, sigma sweep, fuzz slice;
# profiling build: K1z's ablation table; counters build: K1z's tile classes.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
cd $R
OUTNAME=r06/pmc bash tools/pmc_k1.sh
cd $R; cp $O/pmc/summary.txt $O/pmc_summary.txt
python tools/hbm_traffic.py $O/pmc "${COMMIT:-unknown}" > $O/hbm_traffic.json
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
EDHIP_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2> $O/bench_2r.err | grep '^{' > $O/bench_cfg5_2ranks_gloo.json
EDHIP_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 --collective 2> $O/bench_coll.err | grep '^{' > $O/bench_cfg5_collective_gloo.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r06 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --repeats 0 > $O/prof.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof/r06_kernel_stats.csv > $O/kernel_stats.txt
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
{ for s in 5 7.5 10 12.5 15 20; do python tools/time_fwd.py $s 2>&1 | grep -v amdgpu; python tools/time_grad.py $s 2>&1 | grep -v amdgpu; done; } > $O/sigma_sweep.txt
# ---- fuzz campaign on this build
{ timeout 600 python tests/fuzz/fuzz_hot.py 9101 300; timeout 600 python tests/fuzz/fuzz_parity.py 9102 200; timeout 300 python tests/fuzz/fuzz_int.py 9103 150;
  timeout 600 python tests/fuzz/fuzz_round4.py 9104 150; timeout 300 python tests/fuzz/fuzz_filter.py 9105 300; timeout 300 python tests/fuzz/fuzz_api.py 9106 150; } 2>&1 | grep -v amdgpu.ids | grep "cases\|FAIL" > $O/fuzz.txt
# ---- profiling build
cp elasticdeform_amd/libedhip.so /tmp/libedhip_ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
echo "# K1 (k1z_tile_kernel<3, false>, profiling build): level-1 launch with parts of the kernel switched off (EDHIP_TILE_DBG bits) or configured differently; tools/time_k12.py"
TAG="full                                   " ITERS=30 T 256 3 5
TAG="no gather (1<<17)                      " EDHIP_TILE_DBG=131072 ITERS=30 T 256 3 5
TAG="no staging (1<<18)                     " EDHIP_TILE_DBG=262144 ITERS=30 T 256 3 5
TAG="no stores (1<<20)                      " EDHIP_TILE_DBG=1048576 ITERS=30 T 256 3 5
TAG="no gather, no staging                  " EDHIP_TILE_DBG=393216 ITERS=30 T 256 3 5
TAG="no gather, no staging, no stores       " EDHIP_TILE_DBG=1441792 ITERS=30 T 256 3 5
TAG="large boxes (52 KiB, 3 workgroups / CU)" EDHIP_LARGE_BOXES=2 ITERS=30 T 256 3 5
TAG="strips of 8 tiles                      " EDHIP_ZSTRIP=8 ITERS=30 T 256 3 5
TAG="strips of 2 tiles                      " EDHIP_ZSTRIP=2 ITERS=30 T 256 3 5
TAG="round-5 kernel (deform_k1.hip)         " EDHIP_K1_R5=1 ITERS=30 T 256 3 5
TAG="order 1                                " ITERS=30 T 256 1 5
TAG="order 1, round-5 kernel                " EDHIP_K1_R5=1 ITERS=30 T 256 1 5
} > $O/ablate_k1.txt 2>&1
{ for s in 5 10; do python tools/geo_phases.py $s 2>&1 | grep -v amdgpu; done; } > $O/geo_phases.txt
cp tools/libedhip_stats.so elasticdeform_amd/libedhip.so
{
echo "# K1z tile classes and redone windows (counters build, tools/k1_stats.py: side order sigma [control points] [mode])"
for a in "256 3 5" "256 3 10" "256 3 15" "256 3 5 8" "256 3 5 3" "256 1 5" "256 2 5" "256 3 5 5 constant" "256 3 5 5 nearest" "256 3 5 5 reflect" "256 3 5 5 wrap" "128 3 5" "200 3 5" "64 3 3 13"; do
  timeout 100 python tools/k1_stats.py $a 2>&1 | grep k1z | tail -1
done
} > $O/k1_stats.txt 2>&1
cp /tmp/libedhip_ship.so elasticdeform_amd/libedhip.so
tail -c 1500 $O/bench_cfg2.json; echo; cat $O/fuzz.txt; cat $O/ablate_k1.txt
