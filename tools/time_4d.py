#!/usr/bin/env python3
"""dev tool: four deformed axes, deform_grid / deform_grid_gradient wall time per call"""
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


for shape in ((32,) * 4, (16, 32, 32, 64), (48,) * 4):
    for dt in (np.float32, np.float64):
        for order, pf in ((3, True), (3, False), (1, False), (0, False)):
            X = torch.from_numpy(rng.random(shape).astype(dt)).to(dev)
            d = torch.from_numpy(rng.standard_normal((4, 3, 3, 3, 3)) * 3.0).to(dev)
            t = timeit(lambda: ed.deform_grid(X, d, order=order, mode="mirror", prefilter=pf))
            line = "4d %-16s %-8s order %d prefilter %d fwd %7.3f ms %6.0f Mvox/s" % (shape, np.dtype(dt).name, order, pf, t, np.prod(shape) / t / 1e3)
            if shape == (32,) * 4 and pf is False or order == 3 and pf:
                tg = timeit(lambda: ed.deform_grid_gradient(X, d, order=order, mode="mirror", prefilter=pf), 5)
                line += "   grad %7.3f ms" % tg
            print(line, flush=True)
