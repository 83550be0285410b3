#!/bin/bash
# round 5: grid prefilter inside the tables launch, A/B on one box (profiling build)
cd /root/repo; O=gpurun_out/r05p; mkdir -p $O
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
for rep in 1 2; do
for sep in 0 1; do
  if [ $sep = 1 ]; then export EDHIP_GRIDPF_SEPARATE=1; else unset EDHIP_GRIDPF_SEPARATE; fi
  echo "separate=$sep" >> $O/ab.txt
  python tools/time_k12.py 256 3 5 2>/dev/null >> $O/ab.txt
  python tools/time_k12.py 64 3 5 2>/dev/null >> $O/ab.txt
  python bench.py --no-cpu-baseline --no-stress 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench step', d['ms_per_step'], 'K2 call', d['roofline']['whole_call_avg_us'], 'K1 call', d['north_star_kernel']['whole_call_avg_us'])" >> $O/ab.txt
  python tools/latency_small.py 2>/dev/null | grep "autograd fwd+bwd  \|forward  " >> $O/ab.txt
done; done
unset EDHIP_GRIDPF_SEPARATE
cp /tmp/ship.so elasticdeform_amd/libedhip.so
cat $O/ab.txt
