#!/usr/bin/env python3
"""dev tool: where the HOST time of a small call goes (cProfile over 2000 calls of cfg1 and a 32^3 autograd round trip)."""
import cProfile, pstats, io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
import elasticdeform_amd.torch as et
dev = torch.device("cuda", 0)
X1 = torch.zeros((200, 300), device=dev); X1[::10, ::10] = 1
d1 = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 3, 3)) * 25).to(dev)
n = 32
X = torch.rand((n, n, n), device=dev).requires_grad_()
d = torch.from_numpy(np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * 2.5 * n / 128).to(dev)
dY = torch.rand((n, n, n), device=dev)
def rt():
    y = et.deform_grid(X, d, order=3, mode="mirror"); y.backward(dY); X.grad = None
def c1():
    ed.deform_grid(X1, d1, order=3)
for name, fn, N in (("cfg1", c1, 2000), ("rt32", rt, 1000)):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    torch.cuda.synchronize()
    print("%s: %.1f us per call (wall, GPU queue drained at the end)" % (name, (time.perf_counter() - t0) / N * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N): fn()
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print("\n".join(l[:150] for l in s.getvalue().split("\n")[:40]))
