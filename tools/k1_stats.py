#!/usr/bin/env python3
"""dev tool (profiling build copied over elasticdeform_amd/libedhip.so): K1's tile classes and fix-up counters on the
bench field -- how often a window falls outside its sampled box.  python tools/k1_stats.py [side] [order] [sigma] [ncp]"""
import ctypes
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
ncp = int(sys.argv[4]) if len(sys.argv) > 4 else 5
mode = sys.argv[5] if len(sys.argv) > 5 else "mirror"
L = _lib.load()
L.edhip_debug_k1_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
L.edhip_debug_k1z_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
st = (ctypes.c_ulonglong * 8)()
sz = (ctypes.c_ulonglong * 8)()
dev = torch.device("cuda", 0)
X = torch.from_numpy(np.random.default_rng(2).random((n, n, n), dtype=np.float32)).to(dev)
disp = np.random.default_rng(22).standard_normal((3, ncp, ncp, ncp)) * (sigma * n / 256)
L.edhip_debug_k1_stats(st)
L.edhip_debug_k1z_stats(sz)
y = ed.deform_grid(X, disp, order=order, mode=mode, prefilter=False)
torch.cuda.synchronize()
L.edhip_debug_k1_stats(st)
L.edhip_debug_k1z_stats(sz)
if sz[3] + sz[4] + sz[5] + sz[6]:
    print("n=%d order=%d sigma=%g ncp=%d %s (k1z): class-A tiles %d, general %d, taken as two z halves %d, unfit %d; strips on the "
          "general kernel's list %d, waves in the fix-up for a miss %d, voxels redone %d, unfit voxels %d"
          % (n, order, sigma, ncp, mode, sz[3], sz[4], sz[6], sz[5], sz[7], sz[0], sz[1], sz[2]))
print("n=%d order=%d sigma=%g ncp=%d %s: fast tiles %d, general %d, unfit %d; waves with a miss %d, voxels redone %d, unfit voxels %d"
      % (n, order, sigma, ncp, mode, st[3], st[4], st[5], st[0], st[1], st[2]))
