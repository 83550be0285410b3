#!/usr/bin/env python3
"""dev tool: variants of a dumped fuzz_hot failure (FUZZ_DUMP=file.npz).  python tools/repro_case.py file.npz"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
from oracle import ed_oracle as orc
d = np.load(sys.argv[1])
X, dY = d["X"], d["dY"]
full = X.shape
axis = (1, 2, 3) if X.ndim == 4 else None
dev = torch.device("cuda", 0)
def run(tag, order, mode, stale, channels=True, repeat=1):
    x = X if channels else X[0]
    dy = dY if channels else dY[0]
    kw = dict(order=order, mode=mode, cval=float(d["cval"]), prefilter=bool(d["prefilter"]))
    if channels and axis:
        kw["axis"] = axis
    dd = torch.from_numpy(d["disp_fwd"] if stale else d["disp"]).to(dev)
    ed.deform_grid(torch.from_numpy(x).to(dev), dd, **kw)
    if stale:
        dd.data.copy_(torch.from_numpy(d["disp"]))
    truth = orc.deform_grid_gradient(dy.astype(np.float64), d["disp"], X_shape=x.shape, **kw)
    for r in range(repeat):
        gg = ed.deform_grid_gradient(torch.from_numpy(dy).to(dev), dd, X_shape=x.shape, **kw).cpu().numpy()
        e = np.abs(gg - truth)
        print("%-44s err %.3e at %s (truth there %.4f, got %.4f)" % (tag + (" rep %d" % r if repeat > 1 else ""), e.max(), np.unravel_index(e.argmax(), e.shape), truth.flat[e.argmax()], gg.flat[e.argmax()]))
    return gg, truth
o, m = int(d["order"]), str(d["mode"])
run("as dumped (stale boxes)", o, m, True, repeat=3)
run("fresh boxes", o, m, False)
run("stale, one channel", o, m, True, channels=False)
for oo in (3, 4):
    run("stale, order %d" % oo, oo, m, True)
run("stale, mirror", o, "mirror", True)
g, t = run("as dumped again", o, m, True)
bad = np.argwhere(np.abs(g - t) > 1e-3)
print("bad voxels:", len(bad), "bbox lo", bad.min(0) if len(bad) else None, "hi", bad.max(0) if len(bad) else None)
