#!/bin/bash
# dev tool: PMC passes over tools/time_k12.py (counters only; run on the GPU box)
#   OUTNAME=pmc_hot ABLV=0 bash tools/pmc_hot.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${OUTNAME:-pmc_hot}
rm -rf $OUT; mkdir -p $OUT
run() { # name, counters
  ITERS=6 EDHIP_HOT_ABL=${ABLV:-0} rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o p --output-format csv -- python $R/tools/time_k12.py > $OUT/$1.log 2>&1
}
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
run b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
if [ -n "$PMC_MORE" ]; then
run c "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
run d "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
fi
python $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
