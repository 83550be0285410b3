#!/usr/bin/env python3
"""dev tool: prefilter passes on short lines (the exact kernels: < 64 samples, or integer volumes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
dg = sys.modules["elasticdeform_amd.deform_grid"]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or (16, 32, 48, 63, 100, 128):
    for dt in (np.float32, np.int16):
        X = torch.from_numpy((rng.random((n, n, n)) * 200).astype(dt)).cuda()
        us = timed(lambda: dg._filter_axes(X, (0, 1, 2), 3, False, X.device))
        print("%s%s %3d^3 prefilter, 3 axes: %7.1f us" % (os.environ.get("TAG", ""), np.dtype(dt).name, n, us))
