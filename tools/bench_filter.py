#!/usr/bin/env python3
"""Per-axis timing of edhip_spline_filter1d on a 256^3 float32 volume (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
from elasticdeform_amd import _lib
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", "256"))
dt = torch.float64 if os.environ.get("F64") else torch.float32
x = torch.rand((n, n, n), device=dev, dtype=dt)
y = torch.empty_like(x)
stream = torch.cuda.current_stream(dev).cuda_stream
def run(axis, tr):
    _lib.spline_filter1d(dgm._desc(x), dgm._desc(y), axis, 3, tr, 0, stream)
for tr in ((0, 1) if not os.environ.get("FWD_ONLY") else (0,)):
    for axis in (0, 1, 2):
        for _ in range(3): run(axis, tr)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record(); run(axis, tr); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        print("transpose=%d axis=%d  median %.1f us  min %.1f us" % (tr, axis, ts[10] * 1e3, ts[0] * 1e3))
# plain copy for scale
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in evs:
    a.record(); y.copy_(x); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in evs)
print("torch copy_  median %.1f us  min %.1f us" % (ts[10] * 1e3, ts[0] * 1e3))
