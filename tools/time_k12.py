#!/usr/bin/env python3
"""dev tool: level-1 launch time of K1 (forward) and K2 (gradient) on the bench workload
(256^3 float32, order 3, mirror, 5^3 grid sigma 5), HIP events inside the library.
  python tools/time_k12.py [side] [order] [sigma]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
from elasticdeform_amd import _lib

dgm = importlib.import_module("elasticdeform_amd.deform_grid")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
order = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
dY = torch.from_numpy(np.random.default_rng(7).random((n, n, n), dtype=np.float32)).to(dev)
disp = torch.from_numpy(np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * (sigma * n / 256)).to(dev)
Xf = dgm._filter_axes(X, [0, 1, 2], order, False, dev) if order > 1 else X
df = dgm._filter_axes(disp, [1, 2, 3], 3, False, dev)
out = torch.empty_like(X)
dxs = torch.zeros_like(X)
stream = torch.cuda.current_stream(dev).cuda_stream
mode = 3
boxes = os.environ.get("BOXES", "1") != "0"      # forward -> gradient hand-over of the tile boxes (the step's normal mode)
args_f = ([dgm._desc(Xf)], dgm._desc(df), None, [dgm._desc(out)], [(0, 1, 2)], [order], [mode], [0.0], None,
          _lib.FLAG_AUTO | (_lib.FLAG_KEEP_BOXES if boxes else 0), stream)
args_g = ([dgm._desc(dxs)], dgm._desc(df), None, [dgm._desc(dY)], [(0, 1, 2)], [order], [mode], [0.0], None,
          _lib.FLAG_AUTO | (_lib.FLAG_USE_BOXES if boxes else 0), stream)
L = _lib.load()


def dom(grad, args, iters=int(os.environ.get("ITERS", "40"))):
    for _ in range(3):
        _lib.deform(grad, *args)
    torch.cuda.synchronize()
    L.edhip_profile_dominant(1)
    ts = []
    for _ in range(iters):
        _lib.deform(grad, *args)
        us = L.edhip_profile_last_us()
        if us > 0:
            ts.append(us)
    L.edhip_profile_dominant(0)
    ts.sort()
    return ts[len(ts) // 2] if ts else -1


def whole(grad, args, iters=int(os.environ.get("ITERS", "40"))):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        _lib.deform(grad, *args)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


tag = os.environ.get("TAG", "")
print("%s n=%d order=%d sigma=%g  K1 level1 %.1f us (call %.1f)   K2 level1 %.1f us (call %.1f)" %
      (tag, n, order, sigma, dom(False, args_f), whole(False, args_f), dom(True, args_g), whole(True, args_g)), flush=True)
