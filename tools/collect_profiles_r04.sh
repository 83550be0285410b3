#!/bin/bash
# Final collection of round 4: COMMIT=$(git rev-parse --short HEAD) gpurun -- "COMMIT=$COMMIT bash tools/collect_profiles_r04.sh"
# shipped library: bench lines (cfg2 with fresh_grid / stress, cfg4, cfg5, 2 ranks on the one GPU, collective leg), rocprofv3
# kernel stats of the same bench command, PMC passes + HBM traffic, time matrix; profiling build: the records route
# (hot_grad2_kernel) against the shipped gradient kernel -- times, per-wave phase clocks, PMC.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
# PMC passes first: bench.py reports roofline.traffic from profiles/hbm_traffic.json when its source hash matches
cd $R
OUTNAME=r04/pmc bash tools/pmc_hot.sh
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ITERS=6 rocprofv3 --kernel-trace --pmc $c -d $O/pmc/$c -o p --output-format csv -- python $R/tools/time_k12.py > $O/pmc/$c.log 2>&1
done
cd $R; python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/hbm_traffic.py $O/pmc "${COMMIT:-unknown}" > $O/hbm_traffic.json
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
EDHIP_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_2ranks_gloo.json 2> $O/bench_2r.err
EDHIP_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload cfg5 --batch 8 --steps 5 --warmup 2 --collective > $O/bench_cfg5_collective_gloo.json 2> $O/bench_coll.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o r04 --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress > $O/prof.log 2>&1
cd $R; python tools/kernel_stats_csv.py $O/prof/r04_kernel_stats.csv > $O/kernel_stats.txt
cd /tmp
for w in auto off; do
  WINDOW=$w rocprofv3 --kernel-trace --stats -d $O/cfg4_$w -o p --output-format csv -- python $R/tools/cfg4_calls.py 10 > $O/cfg4_$w.log 2>&1
  python $R/tools/kernel_stats_csv.py $O/cfg4_$w/p_kernel_stats.csv > $O/cfg4_stats_$w.txt 2>/dev/null
done
cd $R
T() { timeout 120 python tools/time_k12.py "$@" 2>&1 | tail -1; }
{ for o in 1 2 3; do T 256 $o 5; done; T 256 3 10; T 256 3 15; T 128 3 5; } > $O/misc.txt 2>/dev/null
timeout 600 python tools/time_matrix.py 2>&1 | grep -v amdgpu.ids > $O/time_matrix.txt
timeout 300 python tools/time_crop_window.py 2>&1 | grep -v amdgpu.ids > $O/time_crop_window.txt
if [ -n "$SKIP_RECORDS" ]; then tail -c 900 $O/bench_cfg2.json; exit 0; fi
# ---- profiling build: records route vs shipped route
cp elasticdeform_amd/libedhip.so /tmp/libedhip_ship.so
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
{
echo "# hot_grad2_kernel (records route, EDHIP_RECORDS=1, profiling build) against hot_grad_kernel, tools/time_k12.py"
for s in 5 10; do
  TAG="records sigma $s" EDHIP_RECORDS=1 ITERS=20 timeout 200 python tools/time_k12.py 256 3 $s
  TAG="shipped sigma $s" ITERS=20 timeout 200 python tools/time_k12.py 256 3 $s
done
TAG="records, no flush (dbg 4)     " EDHIP_RECORDS=1 EDHIP_TILE_DBG=4 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="records, no consumers (dbg 8) " EDHIP_RECORDS=1 EDHIP_TILE_DBG=8 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="records, neither (dbg 12)     " EDHIP_RECORDS=1 EDHIP_TILE_DBG=12 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="records, standalone gradient  " EDHIP_RECORDS=1 BOXES=0 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
TAG="shipped, standalone gradient  " BOXES=0 ITERS=20 timeout 200 python tools/time_k12.py 256 3 5
EDHIP_RECORDS=1 timeout 200 python tools/g2_phases.py 5
EDHIP_RECORDS=1 timeout 200 python tools/g2_phases.py 10
} 2>&1 | grep -v amdgpu.ids > $O/records_route.txt
EDHIP_RECORDS=1 OUTNAME=r04/pmc_records bash tools/pmc_hot.sh
cp /tmp/libedhip_ship.so elasticdeform_amd/libedhip.so
tail -c 900 $O/bench_cfg2.json; echo; tail -c 400 $O/bench_cfg4.json; echo; cat $O/records_route.txt | head -12
