#!/usr/bin/env python3
"""dev tool: the benchmark step (deform_grid + deform_grid_gradient, 256^3 float32, 5^3 grid, order 3, mirror, prefilter on)
in a loop, for rocprofv3 --kernel-trace.   python tools/prof_step.py [sigma] [iters]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 256
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
dY = torch.from_numpy(np.random.default_rng(3).random((n, n, n), dtype=np.float32)).to(dev)
d = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sigma).to(dev)
for _ in range(iters):
    ed.deform_grid(X, d, order=3, mode="mirror")
    ed.deform_grid_gradient(dY, d, order=3, mode="mirror")
torch.cuda.synchronize()
