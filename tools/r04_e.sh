#!/bin/bash
# PMC passes over time_k12.py (new records route and old route), profiling build
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd $R
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
PMC_MORE=1 OUTNAME=r04e_new bash tools/pmc_hot.sh
EDHIP_NO_RECORDS=1 OUTNAME=r04e_old bash tools/pmc_hot.sh
grep -A40 "hot_grad" gpurun_out/r04e_new/summary.txt | grep -v "^==" | head -120
echo ======= OLD
grep -A12 "hot_grad_kernel" gpurun_out/r04e_old/summary.txt | head -80
