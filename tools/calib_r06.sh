# dev tool (round 6): FETCH_SIZE / WRITE_SIZE / L2 hit counters on the access patterns of K1z (tools/ubench_traffic.bin) and on
# K1z / K2 themselves (tools/time_k12.py), one counter group per pass.   gpurun -- 'bash tools/calib_r06.sh'
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06cal; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_HIT[A-Za-z0-9_]*\|TCC_MISS[A-Za-z0-9_]*\|TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_READ[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $O/counters.txt
pass() { # name, counters, command...
  local n=$1 c=$2; shift 2
  rocprofv3 --kernel-trace --pmc $c -d $O/$n -o p --output-format csv -- "$@" > $O/$n.log 2>&1 || echo "pass $n failed" >> $O/failed.txt
}
for t in ub k; do
  if [ $t = ub ]; then CMD="$R/tools/ubench_traffic.bin"; else CMD="python $R/tools/time_k12.py"; export ITERS=6; fi
  pass ${t}_f "FETCH_SIZE" $CMD
  pass ${t}_w "WRITE_SIZE" $CMD
  pass ${t}_h "TCC_HIT_sum TCC_MISS_sum" $CMD
  pass ${t}_r "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" $CMD
  pass ${t}_q "TCC_REQ_sum TCC_READ_sum" $CMD
done
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r06cal'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
with open(O + '/summary.txt', 'w') as out:
    for k, cs in agg.items():
        if not any(t in k for t in ('rd4_', 'rd16_', 'lds16_', 'wr4_', 'k1z_tile', 'hot_grad', 'k1z_fix', 'k1z_geo')):
            continue
        out.write(k[:110] + '\n')
        for c, v in sorted(cs.items()):
            out.write('    %-24s n=%-3d mean=%.6g\n' % (c, len(v), sum(v) / len(v)))
print(open(O + '/summary.txt').read())
PY
cat $O/counters.txt; cat $O/failed.txt 2>/dev/null
