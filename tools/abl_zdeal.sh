# dev tool (profiling build): K1z forward call and its fabric reads against the XCD dealing of the strips (EDHIP_ZDEAL) and the strip length
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/zdeal; rm -rf $O; mkdir -p $O
for st in 4 8; do for dl in 32 128 512 1024; do
  echo -n "ZSTRIP=$st ZDEAL=$dl "; EDHIP_ZSTRIP=$st EDHIP_ZDEAL=$dl python tools/time_fwd.py 5 2>&1 | grep -v amdgpu
  (cd /tmp && EDHIP_ZSTRIP=$st EDHIP_ZDEAL=$dl ITERS=4 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum -d $O/s${st}d$dl -o p --output-format csv -- python $R/tools/time_fwd.py 5 > /dev/null 2>&1)
  python - <<PY
import csv,glob,collections
v=collections.defaultdict(list)
for f in glob.glob("$O/s${st}d$dl/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k1z_tile' in r['Kernel_Name']: v[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(x)/len(x) for k,x in v.items()}
if m: print("    reads %.1f MB, L2 hit %.3f" % (128*m['TCC_EA0_RDREQ_128B_sum']/1e6, m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum'])))
PY
done; done
