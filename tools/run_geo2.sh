# dev tool (profiling build): the geometry kernel launched twice in a row (EDHIP_GEO_TWICE): cold against warm
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
export TMPDIR=/tmp; cd /tmp && EDHIP_GEO_TWICE=1 rocprofv3 --kernel-trace -d /tmp/g2 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/time_fwd.py 5 > /dev/null 2>&1
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/tmp/g2/p_kernel_trace.csv'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(r['Kernel_Name'][:40], int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows]
first=[];second=[]
for i in range(1,len(d)):
    if 'k1z_geo' in d[i][0] and 'k1z_geo' in d[i-1][0]:
        first.append(d[i-1][1]); second.append(d[i][1])
import statistics as st
print("geometry kernel, first of a pair: median %.1f us; second (warm): median %.1f us; pairs %d" % (st.median(first)/1e3, st.median(second)/1e3, len(first)))
PY
