#!/usr/bin/env python3
"""dev tool: N repeats of one small case, for rocprofv3 --kernel-trace --stats.  usage: small_case.py {cfg1|rt32|rt64|rt128|fw32} [N]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed
import elasticdeform_amd.torch as et
dev = torch.device("cuda", 0)
case = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if case == "cfg1":
    X1 = torch.zeros((200, 300), device=dev); X1[::10, ::10] = 1
    d1 = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 3, 3)) * 25).to(dev)
    fn = lambda: ed.deform_grid(X1, d1, order=3)
else:
    n = int(case[2:])
    X = torch.rand((n, n, n), device=dev).requires_grad_()
    d = torch.from_numpy(np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * 2.5 * n / 128).to(dev)
    dY = torch.rand((n, n, n), device=dev)
    if case.startswith("rt"):
        def fn():
            y = et.deform_grid(X, d, order=3, mode="mirror"); y.backward(dY); X.grad = None
    else:
        fn = lambda: ed.deform_grid(X.detach(), d, order=3, mode="mirror")
for _ in range(N):
    fn()
torch.cuda.synchronize()
