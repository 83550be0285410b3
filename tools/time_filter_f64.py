#!/usr/bin/env python3
"""dev tool: float64 prefilter chains (all axes), orders 3-5, default arithmetic against the exact kernel"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elasticdeform_amd as ed  # noqa
from elasticdeform_amd import _lib

dgm = importlib.import_module("elasticdeform_amd.deform_grid")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
for shape in ((128, 128, 128), (256, 256, 256), (2048, 2048)):
    for dt in (np.float64, np.float32):
        x = torch.from_numpy(np.random.default_rng(1).random(shape).astype(dt)).to(dev)
        for order in (3, 4, 5):
            for tr in (0, 1):
                line = "%-16s %-8s order %d %s" % ("x".join(map(str, shape)), np.dtype(dt).name, order, "transposed" if tr else "forward   ")
                for name, flag in (("default", _lib.FLAG_AUTO), ("exact", _lib.FLAG_EXACT)):
                    out = torch.empty_like(x)
                    def chain():
                        src = x
                        for ax in range(len(shape)):
                            _lib.spline_filter1d(dgm._desc(src), dgm._desc(out), ax, order, tr, flag, stream)
                            src = out
                    for _ in range(3):
                        chain()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(10):
                        chain()
                    b.record()
                    torch.cuda.synchronize()
                    line += "   %s %8.1f us" % (name, a.elapsed_time(b) * 100)
                print(line, flush=True)
