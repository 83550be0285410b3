# Verdict item 2/3 (sliding register window along x): how often can a lane / a whole wave keep
# (start_z, start_y) and advance start_x by exactly one between x-consecutive voxels?
import numpy as np
n=256
for s in (5,10):
    c = np.load(f'/tmp/sim/coords_s{s}.npy').astype(np.float64)
    c = np.abs(c); c = np.where(c>n-1, 2*(n-1)-c, c)
    st = np.floor(c).astype(np.int32)-1
    d = st[:,:,:,1:] - st[:,:,:,:-1]           # transition between x-neighbours
    same_zy = (d[0]==0)&(d[1]==0)
    reg = same_zy & (d[2]==1)
    print(f'sigma {s}: lane-steps with (dz,dy)=(0,0): {same_zy.mean():.3f}; with (0,0,+1): {reg.mean():.3f}; dx distribution', {k:round(float((d[2]==k).mean()),3) for k in (-1,0,1,2,3)})
    # wave = 8z x 8y lanes walking x inside an 8-tile: steps 1..7 of each tile
    r = reg[:, :, :].reshape(n//8,8,n//8,8,n-1)
    allreg = r.all(axis=(1,3))                  # all 64 lanes regular at this x-step
    zy = same_zy.reshape(n//8,8,n//8,8,n-1).all(axis=(1,3))
    print(f'   wave-steps (64 lanes = 8z x 8y) where ALL lanes keep (z,y): {zy.mean():.4f}; all regular (0,0,+1): {allreg.mean():.4f}')
    r2 = reg.reshape(n//4,4,n//4,4,n-1).all(axis=(1,3))
    print(f'   16-lane groups (4z x 4y) all regular: {r2.mean():.4f}')
