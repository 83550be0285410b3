import numpy as np, scipy.ndimage as ndi, time
def field(n=256, sigma=5.0, seed=22):
    disp = np.random.default_rng(seed).standard_normal((3,5,5,5))*sigma
    df = disp.copy()
    for a in range(1,4):
        df = ndi.spline_filter1d(df, 3, axis=a, mode='mirror')
    ax = [np.linspace(0,4,n) for _ in range(3)]
    zz,yy,xx = np.meshgrid(*ax, indexing='ij')
    co = np.stack([zz,yy,xx]).reshape(3,-1)
    out=[]
    for h in range(3):
        d = ndi.map_coordinates(df[h], co, order=3, mode='mirror', prefilter=False).reshape(n,n,n)
        out.append(d)
    d = np.stack(out)
    idx = np.stack(np.meshgrid(*[np.arange(n)]*3, indexing='ij')).astype(np.float64)
    return idx + d
if __name__ == '__main__':
    for s in (5.0, 10.0):
        t=time.time(); c = field(256, s); print(time.time()-t)
        np.save(f'/tmp/sim/coords_s{int(s)}.npy', c.astype(np.float32))
