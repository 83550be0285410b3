#!/usr/bin/env python3
"""dev tool: 3-D volumes with wide control grids (more than 13 points along x: per-strip Q tables on the tile kernels),
float32 and float64.  Wall time per call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa

dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for dt in (np.float32, np.float64):
    for n, pts in ((256, 5), (256, 13), (256, 16), (256, 32), (128, 16), (128, 32)):
        for order in (3, 1):
            if dt == np.float64 and order == 1:
                continue
            X = torch.from_numpy(rng.random((n, n, n)).astype(dt)).to(dev)
            d = torch.from_numpy(rng.standard_normal((3, pts, pts, pts)) * (40.0 / pts)).to(dev)
            t = timeit(lambda: ed.deform_grid(X, d, order=order, mode="mirror", prefilter=False))
            tg = timeit(lambda: ed.deform_grid_gradient(X, d, order=order, mode="mirror", prefilter=False), 5)
            print("%d^3 %s, %2d^3 control points, order %d (no prefilter): fwd %7.3f ms %6.0f Mvox/s   grad %7.3f ms"
                  % (n, np.dtype(dt).name, pts, order, t, n ** 3 / t / 1e3, tg), flush=True)
