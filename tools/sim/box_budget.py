import numpy as np
n=256
for s in (5,10):
    c = np.load(f'/tmp/sim/coords_s{s}.npy').astype(np.float64)
    c = np.abs(c); c = np.where(c>n-1, 2*(n-1)-c, c)
    st = np.floor(c).astype(np.int32)-1
    for shape in [(8,8,8),(4,8,8)]:
        tz,ty,tx = shape
        r = st.reshape(3,n//tz,tz,n//ty,ty,n//tx,tx)
        lo = r.min(axis=(2,4,6)); hi = r.max(axis=(2,4,6))+4
        ext = hi-lo
        P = np.where(ext[2]<=14,16,np.where(ext[2]<=18,20,np.where(ext[2]<=22,24,32)))
        b = ext[0]*(ext[1]*P+2)*4
        print(s, shape, 'mean', b.mean(), 'P16 frac', (P==16).mean(), ' >10K %.3f >10.6K %.3f >12.8K %.3f >13.3K %.3f >16K %.3f >20K %.4f'%tuple((b>t).mean() for t in (10240, 10900, 13100, 13600, 16384, 20480)))
