import numpy as np, sys, itertools
n=256
rng=np.random.default_rng(0)
def degree(key, nb):
    res = np.empty(key.shape[0], np.int32)
    for i in range(key.shape[0]):
        u = np.unique(key[i]); res[i] = np.bincount(u % nb, minlength=nb).max()
    return res
s=int(sys.argv[1]) if len(sys.argv)>1 else 5
c = np.load(f'/tmp/sim/coords_s{s}.npy').astype(np.float64)
c = np.abs(c); c = np.where(c>n-1, 2*(n-1)-c, c)
st = np.floor(c).astype(np.int32)-1
T=8
tl = [(rng.integers(0,n//T), rng.integers(0,n//T), rng.integers(0,n//T)) for _ in range(100)]
def run(name, Pfun, PSfun, lanemap, walk):
    tot=0; cnt=0; byt=0
    for (tz,ty,tx) in tl:
        s3 = st[:, tz*T:(tz+1)*T, ty*T:(ty+1)*T, tx*T:(tx+1)*T]
        lo = s3.reshape(3,-1).min(1); hi = s3.reshape(3,-1).max(1)+4
        ext = hi-lo
        r = s3 - lo[:,None,None,None]
        ex = ext[2] + (ext[2]&1) + 2
        by = ext[1]
        P = Pfun(ex); PS = PSfun(by,P)
        byt += ext[0]*PS*4
        base = r[0]*PS + r[1]*P + (r[2] & ~1)
        insts=[]
        for k in range(T):
            b = base[k] if walk=='z' else base[:,k,:]   # (y,x) or (z,x)
            insts.append(lanemap(b))
        insts=np.array(insts)
        a = np.concatenate([insts + l0*PS + l1*P for l0 in range(4) for l1 in range(4)])
        for g in (a[:, :32], a[:, 32:]):
            tot += degree(g//2,32).sum() + degree((g+2)//2,32).sum() + degree((g+4)//2,32).sum()
        cnt += a.shape[0]
    print(s, name, 'per voxel-wave %.0f cycles (ideal 96)  mean LDS bytes %.0f'%(tot/cnt*16, byt/len(tl)))
yx = lambda b: b.reshape(-1)                      # lane = y*8+x
xy = lambda b: b.T.reshape(-1)
def yx_il(b): return np.concatenate([b[0::2].reshape(-1), b[1::2].reshape(-1)])   # group0 = even rows
def padto(v, m, mod): return v + ((m - v) % mod)
for walk in ('z','y'):
  for Pname,Pf in [('P=ex', lambda ex: ex), ('P=16', lambda ex: 16 if ex<=16 else 20), ('P=12/16', lambda ex: 12 if ex<=12 else (16 if ex<=16 else 20))]:
    for PSname,PSf in [('nopad', lambda by,P: by*P)]:
        for lm,lf in [('yx',yx),('xy',xy),('yx_il',yx_il)]:
            run(f'walk{walk} {Pname} {PSname} {lm}', Pf, PSf, lf, walk)
