# dev tool (round 6): request-size split of the L2's fabric-side read / write requests (128 / 64 / 32 bytes) on the ubench patterns
# and on K1z / K2.   gpurun -- 'bash tools/calib_r06b.sh'
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06calb; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_EA0_ATOMIC[A-Za-z0-9_]*\|TCC_ATOMIC[A-Za-z0-9_]*\|TCC_WRITE[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $O/counters.txt
pass() { local n=$1 c=$2; shift 2
  rocprofv3 --kernel-trace --pmc $c -d $O/$n -o p --output-format csv -- "$@" > $O/$n.log 2>&1 || echo "pass $n failed" >> $O/failed.txt; }
for t in ub k; do
  if [ $t = ub ]; then CMD="$R/tools/ubench_traffic.bin"; else CMD="python $R/tools/time_k12.py"; export ITERS=6; fi
  pass ${t}_r "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" $CMD
  pass ${t}_w "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $CMD
  pass ${t}_a "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum" $CMD
done
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r06calb'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
with open(O + '/summary.txt', 'w') as out:
    for k, cs in agg.items():
        if not any(t in k for t in ('rd4_', 'rd16_', 'lds16_', 'wr4_', 'k1z_tile', 'hot_grad')):
            continue
        out.write(k[:110] + '\n')
        for c, v in sorted(cs.items()):
            out.write('    %-28s n=%-3d mean=%.6g\n' % (c, len(v), sum(v) / len(v)))
print(open(O + '/summary.txt').read())
PY
cat $O/counters.txt; cat $O/failed.txt 2>/dev/null
