#!/usr/bin/env python3
"""dev tool: where the time of an integer volume with order 3 goes (exact prefilter per axis, deform)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
from elasticdeform_amd import _lib
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
def timed(fn, iters=7):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[iters // 2] * 1e3
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for dt in (np.int16, np.uint8, np.int32):
    X = torch.from_numpy((rng.random((n, n, n)) * 200).astype(dt)).to(dev)
    d = torch.from_numpy(rng.standard_normal((3, 5, 5, 5)) * 5 * n / 256).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = torch.empty_like(X)
    for ax in range(3):
        print("%s %d^3 exact prefilter axis %d: %.0f us" % (dt.__name__, n, ax, timed(lambda: _lib.spline_filter1d(dgm._desc(X), dgm._desc(out), ax, 3, 0, _lib.FLAG_AUTO, stream))))
    Xf = dgm._filter_axes(X, [0, 1, 2], 3, False, dev)
    print("%s %d^3 order 3 deform_grid whole call: %.0f us; prefilter=False: %.0f us; order 1: %.0f us; order 0: %.0f us" % (
        dt.__name__, n, timed(lambda: ed.deform_grid(X, d, order=3, mode="mirror")),
        timed(lambda: ed.deform_grid(X, d, order=3, mode="mirror", prefilter=False)),
        timed(lambda: ed.deform_grid(X, d, order=1, mode="mirror")),
        timed(lambda: ed.deform_grid(X, d, order=0, mode="mirror"))))
