#!/usr/bin/env python3
"""dev tool: event-timed gradient call (256^3 float32, 5^3 grid, order 3, mirror, prefilter off) after a forward call with
the same grid (boxes handed over), median of repeats.   python tools/time_grad.py [sigma]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
n = 256
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
dY = torch.from_numpy(np.random.default_rng(3).random((n, n, n), dtype=np.float32)).to(dev)
d = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sigma).to(dev)
for _ in range(6):
    ed.deform_grid(X, d, order=3, mode="mirror", prefilter=False)
    ed.deform_grid_gradient(dY, d, order=3, mode="mirror", prefilter=False)
torch.cuda.synchronize()
ts = []
for rep in range(7):
    ed.deform_grid(X, d, order=3, mode="mirror", prefilter=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ed.deform_grid_gradient(dY, d, order=3, mode="mirror", prefilter=False)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 10)
print("gradient call (prefilter off) sigma %g: median %.1f us, min %.1f, max %.1f" % (sigma, float(np.median(ts)), min(ts), max(ts)))
