#!/usr/bin/env python3
"""dev tool: throughput of deform_grid (and its gradient where defined) over a matrix of shapes / dtypes / orders,
to find the cases that fall far below the float32 3-D tile path.  Prints Mvoxels/s per call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def tensor(shape, dtype):
    if np.issubdtype(dtype, np.floating):
        return torch.from_numpy(rng.random(shape).astype(dtype)).to(dev)
    return torch.from_numpy(rng.integers(0, 100, shape).astype(dtype)).to(dev)


cases = []
for dt in (np.float32, np.float64, np.uint8, np.int16, np.int32):
    for order in (0, 1, 3, 5):
        cases.append(("2d 1024^2", (1024, 1024), None, dt, order, (2, 5, 5), {}))
        cases.append(("3d 128^3", (128, 128, 128), None, dt, order, (3, 5, 5, 5), {}))
for dt in (np.float32, np.uint8):
    for order in (0, 1, 3):
        cases.append(("2d+ch 8x512^2", (8, 512, 512), (1, 2), dt, order, (2, 5, 5), {}))
        cases.append(("3d+ch 4x96^3", (4, 96, 96, 96), (1, 2, 3), dt, order, (3, 4, 4, 4), {}))
        cases.append(("3d ch-last 96^3x4", (96, 96, 96, 4), (0, 1, 2), dt, order, (3, 4, 4, 4), {}))
cases.append(("4d 32^4", (32, 32, 32, 32), None, np.float32, 3, (4, 3, 3, 3, 3), {}))
cases.append(("4d 32^4", (32, 32, 32, 32), None, np.float32, 1, (4, 3, 3, 3, 3), {}))
cases.append(("1d 1M", (1 << 20,), None, np.float32, 3, (1, 8), {}))
for name, shape, axis, dt, order, dshape, kw in cases:
    X = tensor(shape, dt)
    d = torch.from_numpy(rng.standard_normal(dshape) * 3.0).to(dev)
    n = int(np.prod(shape))
    try:
        t = timeit(lambda: ed.deform_grid(X, d, order=order, mode="mirror", axis=axis, **kw))
        line = "%-20s %-8s order %d  fwd %8.3f ms %9.0f Mvox/s" % (name, np.dtype(dt).name, order, t, n / t / 1e3)
        if np.issubdtype(dt, np.floating):
            dY = tensor(shape, dt)
            tg = timeit(lambda: ed.deform_grid_gradient(dY, d, order=order, mode="mirror", axis=axis, **kw))
            line += "   grad %8.3f ms %9.0f Mvox/s" % (tg, n / tg / 1e3)
        print(line, flush=True)
    except Exception as e:      # noqa
        print("%-20s %-8s order %d  FAILED %s" % (name, np.dtype(dt).name, order, e), flush=True)
