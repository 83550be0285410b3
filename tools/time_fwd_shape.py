#!/usr/bin/env python3
"""dev tool: event-timed forward call (float32, 5^3 grid sigma 5 * z/256, order 3, mirror, prefilter off) for a list of shapes.
python tools/time_fwd_shape.py 256x256x264 320x320x256 ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    shape = tuple(int(v) for v in arg.split("x"))
    X = torch.rand(shape, device=dev)
    sig = np.array([float(os.environ.get("SIGMA", "5")) * s / 256 for s in shape[-3:]]).reshape(3, 1, 1, 1)
    d = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sig).to(dev)
    fn = ed.deform_grid_batch if len(shape) == 4 else ed.deform_grid
    if len(shape) == 4:
        d = d[None].repeat(shape[0], 1, 1, 1, 1).contiguous()
    for _ in range(8):
        fn(X, d, order=3, mode="mirror", prefilter=False)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(X, d, order=3, mode="mirror", prefilter=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 10)
    vox = float(np.prod(shape))
    print("%-16s forward call %8.1f us   %6.2f ns per 1000 voxels" % (arg, float(np.median(ts)), float(np.median(ts)) * 1e3 / vox * 1e3 / 1e3))
    del X
