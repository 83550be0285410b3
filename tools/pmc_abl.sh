#!/bin/bash
# instruction counts of compile-time ablated builds of the K1 tile kernel
cd /tmp && export TMPDIR=/tmp
for d in 0 2 4 6 16 38; do
  OUT=/tmp/pmcabl$d; rm -rf $OUT
  EDHIP_TILE_DBG=$d rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -d $OUT -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== ABL=$d"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | grep -A8 "fwd_kernel" | grep -v "^   void" | awk '{printf "%s %s | ", $1, $3}'; echo
  EDHIP_TILE_DBG=$d timeout 100 python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   K1_forward ms', d['phases_ms']['K1_forward'])"
done
