#!/bin/bash
# dev tool: PMC passes over tools/time_k12.py for the K1 / K2 kernels, incl. GRBM_GUI_ACTIVE (busy cycles of the chip:
# VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / (GRBM_GUI_ACTIVE per SE ...), see profiles/r05_pmc_summary.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${OUTNAME:-pmc_k1}
rm -rf $OUT; mkdir -p $OUT
run() { ITERS=6 rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o p --output-format csv -- python $R/tools/time_k12.py > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
run b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
run g "GRBM_GUI_ACTIVE GRBM_COUNT"
# fabric-side bytes of the L2s (profiles/r06_traffic_calibration.txt): requests by size, not FETCH_SIZE (= requests x 64 B whatever their size)
run r "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
run w "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
run h "TCC_HIT_sum TCC_MISS_sum"
run f "FETCH_SIZE"
python $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
