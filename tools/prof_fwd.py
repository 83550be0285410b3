#!/usr/bin/env python3
"""dev tool: the benchmark's forward call (256^3 float32, 5^3 grid, order 3, mirror, prefilter off) in a loop, for
rocprofv3 --kernel-trace --stats.   python tools/prof_fwd.py [sigma] [iters] [side] [order]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
order = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
d = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sigma * n / 256).to(dev)
for _ in range(iters):
    ed.deform_grid(X, d, order=order, mode="mirror", prefilter=False)
torch.cuda.synchronize()
