import numpy as np, sys
n=256
rng=np.random.default_rng(0)
def degree(addr, nb, unit):
    key = addr//unit
    res = np.empty(addr.shape[0], np.int32)
    for i in range(addr.shape[0]):
        u = np.unique(key[i]); res[i] = np.bincount(u % nb, minlength=nb).max()
    return res
def sim(s, tiles=120):
    c = np.load(f'/tmp/sim/coords_s{s}.npy').astype(np.float64)
    c = np.abs(c); c = np.where(c>n-1, 2*(n-1)-c, c)
    st = np.floor(c).astype(np.int32)-1
    T=8
    tl = [(rng.integers(0,n//T), rng.integers(0,n//T), rng.integers(0,n//T)) for _ in range(tiles)]
    for name in ['M1_tight','M1_2copy','M2_odd2','M2_plane','M2_plane_2copy','M2b_plane']:
        tot=0; cnt=0; byt=0
        for (tz,ty,tx) in tl:
            s3 = st[:, tz*T:(tz+1)*T, ty*T:(ty+1)*T, tx*T:(tx+1)*T]
            lo = s3.reshape(3,-1).min(1); hi = s3.reshape(3,-1).max(1)+4
            ext = hi-lo
            r = s3 - lo[:,None,None,None]
            two = '2copy' in name
            ex = ext[2] + (ext[2]&1) + (0 if two else 2)
            by = ext[1]
            if name.startswith('M1'):
                P = ex if not two else (16 if ex<=16 else 48)
                PS = by*P
            else:
                P = ex if (ex//2)%2==1 else ex+2
                PS = by*P
                if 'plane' in name:
                    PS += (16 - PS) % 64
            byt += ext[0]*PS*4*(2 if two else 1)
            xs = r[2] if two else (r[2] & ~1)
            base = r[0]*PS + r[1]*P + xs
            if two:
                # copy select by parity: odd lanes read from copy1 at (x-1) -> aligned; copy1 base offset 56 mod 64 dwords
                cap = ext[0]*PS; cap = cap + ((56 - cap) % 64)
                base = np.where(r[2]&1, base - 1 + cap, base)
            insts=[]
            if name.startswith('M1'):
                for z in range(T): insts.append(base[z].reshape(-1))
            elif name.startswith('M2b'):
                # lanes: 16-lane... group of 32 = (4 z) x (8 y) but z interleaved: lane = y + 8*z ; same as M2. variant: lanes (z,y) with y fastest but group0 = z even, group1 = z odd
                for x in range(T):
                    b = base[:,:,x]
                    b = np.concatenate([b[0::2].reshape(-1), b[1::2].reshape(-1)])
                    insts.append(b)
            else:
                for x in range(T): insts.append(base[:,:,x].reshape(-1))
            insts = np.array(insts)
            a=[]
            for l0 in range(4):
                for l1 in range(4):
                    a.append(insts + l0*PS + l1*P)
            a = np.concatenate(a)
            for g in (a[:, :32], a[:, 32:]):
                tot += degree(g,32,2).sum() + degree(g+2,32,2).sum()
                if not two: tot += degree(g+4,32,2).sum()   # third read as b64 too
            cnt += a.shape[0]
        print(s, name, 'per voxel-wave %.0f cycles (ideal %d)  mean LDS bytes %.0f'%(tot/cnt*16, 64 if two else 96, byt/len(tl)))
for s in (5,10): sim(s)
