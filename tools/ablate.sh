mkdir -p gpurun_out
for d in ${ABL:-0 1 2 3 4 6 7}; do
  echo "DBG=$d"; EDHIP_TILE_DBG=$d timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phases_ms'])"
done
