#!/usr/bin/env python3
"""dev tool: gradient (and forward) of a 3-D volume with a given number of control points, for rocprofv3 --kernel-trace --stats
  python tools/prof_grid.py side points dtype order [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa

n, pts, dt, order = int(sys.argv[1]), int(sys.argv[2]), np.dtype(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
X = torch.from_numpy(rng.random((n, n, n)).astype(dt)).to(dev)
d = torch.from_numpy(rng.standard_normal((3, pts, pts, pts)) * (40.0 / pts)).to(dev)
for fn in (ed.deform_grid, ed.deform_grid_gradient):
    for _ in range(2):
        fn(X, d, order=order, mode="mirror", prefilter=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn(X, d, order=order, mode="mirror", prefilter=False)
    b.record()
    torch.cuda.synchronize()
    print("%d^3 %s %d^3 points order %d %s %.3f ms" % (n, dt.name, pts, order, fn.__name__, a.elapsed_time(b) / iters), flush=True)
