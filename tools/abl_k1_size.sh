# dev tool: K1z against the round-5 K1 by volume size, both in SHIPPED-library builds (tools/libedhip_nok1z.so: make OBJDIR=obj_nok1z
# OUT=../../tools/libedhip_nok1z.so EXTRA=-DEDHIP_NO_K1Z), forward call incl. geometry / tables; and the cfg5 batch
cp elasticdeform_amd/libedhip.so /tmp/ship.so
for lib in /tmp/ship.so tools/libedhip_nok1z.so; do
  cp $lib elasticdeform_amd/libedhip.so; echo "== $lib"
  for n in 64 96 128 160 192 224 256 288 320 384; do ITERS=20 python tools/time_k12.py $n 3 5 2>&1 | tail -1; done
  for n in 128 256; do ITERS=20 python tools/time_k12.py $n 1 5 2>&1 | tail -1; done
  python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --repeats 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'], d['phases_ms'])"
done
cp /tmp/ship.so elasticdeform_amd/libedhip.so
