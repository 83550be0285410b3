#!/usr/bin/env python3
"""dev tool: the 4-D gradient calls of tools/time_4d.py alone (for rocprofv3 --kernel-trace --stats)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
dev = torch.device("cuda", 0); rng = np.random.default_rng(1)
pf = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
dt = np.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else np.float64
X = torch.from_numpy(rng.random((32,) * 4).astype(dt)).to(dev)
d = torch.from_numpy(rng.standard_normal((4, 3, 3, 3, 3)) * 3.0).to(dev)
for _ in range(8):
    ed.deform_grid_gradient(X, d, order=3, mode="mirror", prefilter=pf)
torch.cuda.synchronize()
