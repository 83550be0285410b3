#!/usr/bin/env python3
"""dev tool: host-bound configurations -- cfg1 (2-D 200x300), cfg3 (128^3 autograd round trip)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
import elasticdeform_amd.torch as et
dev = torch.device("cuda", 0)


def wall(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


X1 = torch.zeros((200, 300), device=dev); X1[::10, ::10] = 1
d1 = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 3, 3)) * 25).to(dev)
print("cfg1 200x300 deform_grid: %.1f us" % wall(lambda: ed.deform_grid(X1, d1, order=3), 500))
for n in (32, 64, 128):
    X = torch.rand((n, n, n), device=dev).requires_grad_()
    d = torch.from_numpy(np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * 2.5 * n / 128).to(dev)
    dY = torch.rand((n, n, n), device=dev)

    def rt():
        y = et.deform_grid(X, d, order=3, mode="mirror")
        y.backward(dY)
        X.grad = None
    print("cfg3-like %d^3 autograd fwd+bwd: %.1f us   fwd only %.1f us" %
          (n, wall(rt, 200), wall(lambda: ed.deform_grid(X.detach(), d, order=3, mode="mirror"), 200)))
