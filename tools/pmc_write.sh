#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the bench kernels for one EDHIP_TILE_DBG setting
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcw${EDHIP_TILE_DBG:-0}
rm -rf $OUT; mkdir -p $OUT
for c in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
