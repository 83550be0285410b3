#!/usr/bin/env python3
"""dev tool: HOST time to enqueue one autograd forward + backward of an n^3 volume (the GPU is kept busy behind a long
kernel, nothing blocks on it) and the wall time per iteration with the queue drained -- which side bounds small volumes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
import elasticdeform_amd.torch as et

dev = torch.device("cuda", 0)
big = torch.empty(256 * 1024 * 1024, device=dev)
for n in (32, 48, 64):
    X = torch.rand((n, n, n), device=dev).requires_grad_()
    d = torch.from_numpy(np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * 2.5 * n / 128).to(dev)
    dY = torch.rand((n, n, n), device=dev)

    def rt():
        y = et.deform_grid(X, d, order=3, mode="mirror"); y.backward(dY); X.grad = None
    for _ in range(50):
        rt()
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        for _ in range(30):
            big.normal_()
        t0 = time.perf_counter()
        for _ in range(200):
            rt()
        host = (time.perf_counter() - t0) / 200 * 1e6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            rt()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 1000 * 1e6
        res.append((host, wall))
    print("%d^3 autograd fwd+bwd: host enqueue %s us, wall %s us" % (n, " ".join("%.1f" % h for h, _ in res), " ".join("%.1f" % w for _, w in res)), flush=True)
