#!/bin/bash
# dev tool (GPU box): PMC passes over tools/prof_fwd.py for the forward kernels; summary on stdout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${OUTNAME:-pmc_fwd}
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o p --output-format csv -- python $R/tools/prof_fwd.py ${SIGMA:-5} 6 > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
run b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
run g "GRBM_GUI_ACTIVE GRBM_COUNT"
if [ -n "$TRAFFIC" ]; then
run f "FETCH_SIZE"
run w "WRITE_SIZE"
run h "TCC_HIT_sum TCC_MISS_sum"
fi
python $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
