# K2 (wave per tile, lanes = (z, y) rows walking x): LDS atomic conflicts of the 64 ds_add_u32 per voxel.
# model: 4 groups of 16 lanes, 32 banks; a group costs max over banks of the lanes hitting that bank
import numpy as np, sys
n=256
rng=np.random.default_rng(0)
s=int(sys.argv[1]) if len(sys.argv)>1 else 5
c = np.load(f'/tmp/sim/coords_s{s}.npy').astype(np.float64)
c = np.abs(c); c = np.where(c>n-1, 2*(n-1)-c, c)
st = np.floor(c).astype(np.int32)-1
T=8
tl = [(rng.integers(0,n//T), rng.integers(0,n//T), rng.integers(0,n//T)) for _ in range(150)]
def lanes_2apart(b):   # b[z,y] -> 64 lanes; 16-lane group = rows two apart in z and y
    out=np.empty(64,b.dtype)
    for lane in range(64):
        zz = 2*(lane&3) + ((lane>>4)&1); yy = 2*((lane>>2)&3) + ((lane>>5)&1)
        out[lane]=b[zz,yy]
    return out
def lanes_zy(b): return b.reshape(-1)
def lanes_yz(b): return b.T.reshape(-1)
def run(name, Pfun, PSfun, lm):
    tot=0; cnt=0; byt=0
    for (tz,ty,tx) in tl:
        s3 = st[:, tz*T:(tz+1)*T, ty*T:(ty+1)*T, tx*T:(tx+1)*T]
        lo = s3.reshape(3,-1).min(1); hi = s3.reshape(3,-1).max(1)+4
        ext = hi-lo; r = s3 - lo[:,None,None,None]
        P = Pfun(ext[2]); PS = PSfun(ext[1],P)
        byt += ext[0]*PS*4
        base = r[0]*PS + r[1]*P + r[2]
        for x in range(T):
            a = lm(base[:,:,x])
            for g in range(4):
                grp = a[g*16:(g+1)*16]
                # taps: all 64 offsets share the pattern shifted by a constant -> conflicts identical up to bank rotation,
                # except the l2 (x) offsets which just rotate banks: one evaluation per (l0,l1) suffices
                for l0 in range(4):
                    for l1 in range(4):
                        bk = (grp + l0*PS + l1*P) % 32
                        tot += np.bincount(bk, minlength=32).max()*4   # x4 for the l2 taps
                        cnt += 4
    print(s, name, 'cycles per 16-lane group %.2f -> per voxel-wave %.0f (ideal 256); mean LDS bytes %.0f'%(tot/cnt, tot/cnt*4*64, byt/len(tl)))
odd = lambda e: e | 1
def pad(v,m,mod): return v + ((m - v) % mod)
for lmn,lm in [('2apart',lanes_2apart),('zy',lanes_zy),('yz',lanes_yz)]:
    run(f'{lmn} tight', lambda e:e, lambda by,P:by*P, lm)
    run(f'{lmn} P odd', odd, lambda by,P:by*P, lm)
    run(f'{lmn} P odd PS=8m32', odd, lambda by,P:pad(by*P,8,32), lm)
    run(f'{lmn} P odd PS=4m32', odd, lambda by,P:pad(by*P,4,32), lm)
    run(f'{lmn} P odd PS=16m32', odd, lambda by,P:pad(by*P,16,32), lm)
    run(f'{lmn} P odd PS odd', odd, lambda by,P:odd(by*P), lm)
    run(f'{lmn} P=1m4 PS=8m32', lambda e: pad(e,1,4), lambda by,P:pad(by*P,8,32), lm)
