#!/bin/bash
# PMC passes over the bench (counters only: no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc${EDHIP_TILE_DBG:-0}
rm -rf $OUT; mkdir -p $OUT
run() { # name, counters
  rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1
}
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
run b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"
if [ -z "$PMC_SHORT" ]; then
run c "FETCH_SIZE"
run d "WRITE_SIZE"
run e "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
fi
