#!/usr/bin/env python3
"""dev tool: which output tiles of a dumped fuzz_hot failure produce the wrong gradient (dY masked tile by tile)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
from oracle import ed_oracle as orc
d = np.load(sys.argv[1])
X, dY = d["X"][0], d["dY"][0]
dev = torch.device("cuda", 0)
kw = dict(order=int(d["order"]), mode=str(d["mode"]), cval=float(d["cval"]), prefilter=bool(d["prefilter"]))
dd = torch.from_numpy(d["disp_fwd"]).to(dev)
ed.deform_grid(torch.from_numpy(X).to(dev), dd, **kw)
dd.data.copy_(torch.from_numpy(d["disp"]))
sh = X.shape
for tz in range(0, sh[0], 8):
    for ty in range(0, sh[1], 8):
        for tx in range(0, sh[2], 8):
            m = np.zeros_like(dY)
            m[tz:tz + 8, ty:ty + 8, tx:tx + 8] = dY[tz:tz + 8, ty:ty + 8, tx:tx + 8]
            truth = orc.deform_grid_gradient(m.astype(np.float64), d["disp"], X_shape=sh, **kw)
            gg = ed.deform_grid_gradient(torch.from_numpy(m).to(dev), dd, X_shape=sh, **kw).cpu().numpy()
            e = np.abs(gg - truth).max()
            if e > 1e-4:
                # the forward call's box of this tile cannot be read from here; print the coordinates' range of the tile under both grids
                print("tile", tz // 8, ty // 8, tx // 8, "err %.3e" % e, "sum|truth| %.3f sum|got| %.3f" % (np.abs(truth).sum(), np.abs(gg).sum()))
                # per-voxel: which voxels of the tile
                bad = []
                for z in range(tz, min(tz + 8, sh[0])):
                    for y in range(ty, min(ty + 8, sh[1])):
                        for x in range(tx, min(tx + 8, sh[2])):
                            one = np.zeros_like(dY); one[z, y, x] = 1.0
                            t1 = orc.deform_grid_gradient(one.astype(np.float64), d["disp"], X_shape=sh, **kw)
                            g1 = ed.deform_grid_gradient(torch.from_numpy(one).to(dev), dd, X_shape=sh, **kw).cpu().numpy()
                            if np.abs(g1 - t1).max() > 1e-4:
                                nz = np.argwhere(np.abs(t1) > 1e-9)
                                bad.append(((z, y, x), float(t1.sum()), float(g1.sum()), nz.min(0).tolist(), nz.max(0).tolist()))
                for b in bad[:12]:
                    print("   voxel", b[0], "sum truth %.4f got %.4f  support lo %s hi %s" % (b[1], b[2], b[3], b[4]))
                print("   bad voxels in tile:", len(bad))
