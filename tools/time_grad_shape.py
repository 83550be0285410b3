#!/usr/bin/env python3
"""dev tool: event-timed gradient call after a forward call (float32, 5^3 grid sigma 5 * extent / 256, order 3, mirror, prefilter off)
for a list of shapes: per-voxel cost against the tile counts.  python tools/time_grad_shape.py 256x256x256 264x256x256 ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    shape = tuple(int(v) for v in arg.split("x"))
    X = torch.rand(shape, device=dev); dY = torch.rand(shape, device=dev)
    sig = np.array([float(os.environ.get("SIGMA", "5")) * s / 256 for s in shape]).reshape(3, 1, 1, 1)
    d = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sig).to(dev)
    kw = dict(order=3, mode="mirror", prefilter=False)
    for _ in range(5):
        ed.deform_grid(X, d, **kw); ed.deform_grid_gradient(dY, d, **kw)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        ed.deform_grid(X, d, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ed.deform_grid_gradient(dY, d, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 10)
    t = float(np.median(ts))
    print("%-14s gradient call %7.1f us   %6.2f ns per 1000 voxels" % (arg, t, t * 1e3 / (np.prod(shape) / 1e3)))
    del X, dY
