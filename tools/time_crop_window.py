#!/usr/bin/env python3
"""dev tool: when does the crop-aware prefilter (edhip_source_window + windowed filter passes) pay?  256^3 float32 order 3, crops of several sizes, window forced on / off."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
X = torch.rand((n, n, n), device=dev)
d = torch.from_numpy(np.random.default_rng(1).standard_normal((3, 5, 5, 5)) * 5.0).to(dev)


def wall(fn, iters=(40 if n == 256 else 10)):
    """wall time per call: the best of four batches (one-off events -- an allocation, a lazily loaded kernel --
    land in one batch: a 1 ms 'per call' outlier of the mean turned out to be a single 40 ms event)"""
    for _ in range(5):
        fn()
    best = 1e30
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters // 4):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (iters // 4) * 1e6)
    return best


for c in ((32, 64, 96, 128, 160, 192) if n == 256 else (128, 256, 320)):
    lo = (n - c) // 2
    crop = (slice(lo, lo + c),) * 3
    dY = torch.rand((c, c, c), device=dev)
    res = []
    for thr in (0.0, 1e18):
        dgm.CROP_WINDOW_MIN_SAVING = thr
        dgm.CROP_WINDOW_MAX_FRACTION = 1.0
        f = wall(lambda: ed.deform_grid(X, d, order=3, mode="mirror", crop=crop))
        g = wall(lambda: ed.deform_grid_gradient(dY, d, order=3, mode="mirror", crop=crop, X_shape=(n, n, n)))
        res.append((f, g))
    m = 32
    sav = n ** 3 - min(n, c + 2 * m) ** 3
    print("crop %3d^3 of %d^3 (upper bound of the saving %5.1f M voxels): forward window %6.1f us / whole %6.1f us   gradient window %6.1f / whole %6.1f us"
          % (c, n, sav / 1e6, res[0][0], res[1][0], res[0][1], res[1][1]))
