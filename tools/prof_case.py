#!/usr/bin/env python3
"""dev tool: one deform_grid / deform_grid_gradient case in a loop, for rocprofv3 --kernel-trace --stats
  python tools/prof_case.py "4,96,96,96" "1,2,3" float32 3 [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa

shape = tuple(int(v) for v in sys.argv[1].split(","))
axis = tuple(int(v) for v in sys.argv[2].split(",")) if sys.argv[2] != "-" else None
dt = np.dtype(sys.argv[3])
order = int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
X = torch.from_numpy(rng.random(shape).astype(dt)).to(dev)
dY = torch.from_numpy(rng.random(shape).astype(dt)).to(dev)
na = len(axis) if axis else len(shape)
d = torch.from_numpy(rng.standard_normal((na,) + (4,) * na) * 3.0).to(dev)
for _ in range(iters):
    ed.deform_grid(X, d, order=order, mode="mirror", axis=axis)
torch.cuda.synchronize()
for _ in range(iters):
    ed.deform_grid_gradient(dY, d, order=order, mode="mirror", axis=axis)
torch.cuda.synchronize()
