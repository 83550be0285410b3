#!/usr/bin/env python3
"""dev tool: cfg5-shaped batch (B x 128^3 float32, one 5^3 grid per sample), forward + gradient through
deform_grid_batch / deform_grid_gradient_batch, data resident.  python tools/time_batch.py [B] [n]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
X = torch.rand((B, n, n, n), device=dev)
dY = torch.rand((B, n, n, n), device=dev)
D = torch.randn((B, 3, 5, 5, 5), device=dev, dtype=torch.float64) * (5.0 * n / 256)
kw = dict(order=3, mode="mirror")


def step():
    y = ed.deform_grid_batch(X, D, **kw)
    g = ed.deform_grid_gradient_batch(dY, D, **kw)
    return y, g


for _ in range(3):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
it = 10
for _ in range(it):
    step()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / it
print("%s B=%d n=%d fwd+grad %.3f ms  %.2f Gvox/s" % (os.environ.get("TAG", ""), B, n, ms, B * n ** 3 / ms / 1e6), flush=True)
