#!/usr/bin/env python3
"""Forward / gradient timings over orders, dtypes and dimensionalities (HIP events, data resident)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import elasticdeform_amd as ed

def timed(fn, iters=7):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[iters // 2]

rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(a).cuda()
res = {}
X3 = T(rng.random((256, 256, 256), dtype=np.float32)); d3 = T(rng.standard_normal((3, 5, 5, 5)) * 5)
X3d = X3.double()
X2 = T(rng.random((4096, 4096), dtype=np.float32)); d2 = T(rng.standard_normal((2, 5, 5)) * 20)
X2d = X2.double()
X1 = T(rng.random((1 << 22,), dtype=np.float32)); d1 = T(rng.standard_normal((1, 9)) * 50)
for order in range(6):
    for name, X, d in (("3d_256_f32", X3, d3), ("3d_256_f64", X3d, d3), ("2d_4096_f32", X2, d2), ("2d_4096_f64", X2d, d2), ("1d_4M_f32", X1, d1)):
        for pf in ((True, False) if order > 1 else (False,)):
            k = "%s_o%d%s" % (name, order, "" if pf else "_nopf")
            res[k + "_fwd"] = round(timed(lambda: ed.deform_grid(X, d, order=order, mode="mirror", prefilter=pf)), 3)
            res[k + "_grad"] = round(timed(lambda: ed.deform_grid_gradient(X, d, order=order, mode="mirror", prefilter=pf)), 3)
for k in sorted(res): print("%-34s %8.3f ms" % (k, res[k]))
