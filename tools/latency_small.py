#!/usr/bin/env python3
"""dev tool: wall time per call of small volumes (the host-bound regime), general path against the
repeat-call lane (elasticdeform_amd/_fastlane.py).  VERDICT r2 item 7's cases: cfg1 (200x300 forward),
cfg3-like n^3 float32 through the autograd wrapper, forward + backward."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
import elasticdeform_amd.torch as et
from elasticdeform_amd import _fastlane

dev = torch.device("cuda", 0)


def wall(fn, n):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def gpu_only(fn, n=200):
    """GPU time per call with the queue kept full: events around n calls enqueued back to back after a long
    kernel that lets the host run ahead"""
    big = torch.empty(256 * 1024 * 1024, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for _ in range(40):
        big.normal_()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


X1 = torch.zeros((200, 300), device=dev); X1[::10, ::10] = 1
d1 = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 3, 3)) * 25).to(dev)
cases = [("cfg1 200x300 deform_grid", lambda: ed.deform_grid(X1, d1, order=3), 3000)]
for n in (32, 64, 128):
    X = torch.rand((n, n, n), device=dev).requires_grad_()
    d = torch.from_numpy(np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * 2.5 * n / 128).to(dev)
    dY = torch.rand((n, n, n), device=dev)

    def rt(X=X, d=d, dY=dY):
        y = et.deform_grid(X, d, order=3, mode="mirror"); y.backward(dY); X.grad = None

    def fw(X=X, d=d):
        ed.deform_grid(X.detach(), d, order=3, mode="mirror")

    def crop(X=X, d=d, n=n):
        c = (slice(n // 4, 3 * n // 4),) * 3
        y = et.deform_grid(X, d, order=3, mode="mirror", crop=c); y.backward(torch.ones_like(y)); X.grad = None
    cases += [("%d^3 autograd fwd+bwd" % n, rt, 1500), ("%d^3 forward" % n, fw, 2000),
              ("%d^3 autograd fwd+bwd, crop to half" % n, crop, 1000)]
for name, fn, n in cases:
    _fastlane.enabled = False
    t_gen = wall(fn, n)
    _fastlane.enabled = True
    t_lane = wall(fn, n)
    g = gpu_only(fn)
    print("%-40s general path %6.1f us   repeat-call lane %6.1f us   (GPU time of the launches %5.1f us)" % (name, t_gen, t_lane, g))

# the same forward + gradient pair (library level, no autograd) captured once in a HIP graph and replayed
print()
for n in (32, 64, 128):
    x = torch.rand((n, n, n), device=dev)
    dy = torch.rand((n, n, n), device=dev)
    d = torch.from_numpy(np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * 2.5 * n / 128).to(dev)
    kw = dict(order=3, mode="mirror")

    def pair():
        ed.deform_grid(x, d, **kw)
        ed.deform_grid_gradient(dy, d, **kw)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            pair()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        pair()
    t_eager = wall(pair, 1500)
    t_graph = wall(g.replay, 1500)
    print("%d^3 deform_grid + deform_grid_gradient: eager (repeat-call lane) %6.1f us   HIP-graph replay %6.1f us" % (n, t_eager, t_graph))
