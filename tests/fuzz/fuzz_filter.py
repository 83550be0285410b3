#!/usr/bin/env python3
"""Randomised sweep of edhip_spline_filter1d against SciPy (forward) and the oracle (transpose):
shapes, axes, orders, dtypes, strided / transposed views, in place and out of place, both
arithmetic modes.  python tests/fuzz/fuzz_filter.py [seed] [cases]"""
import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scipy.ndimage, torch
from elasticdeform_amd import _lib
from oracle import ed_oracle as orc
dgm = importlib.import_module("elasticdeform_amd.deform_grid")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
fails = 0
for case in range(ncases):
    nd = int(rng.integers(1, 4))
    lim = {1: 5000, 2: 700, 3: 150}[nd]
    shape = tuple(int(rng.integers(1, lim)) for _ in range(nd))
    axis = int(rng.integers(0, nd))
    order = int(rng.integers(0, 6))
    dtype = rng.choice([np.float32, np.float32, np.float64, np.int16, np.uint8])
    transpose = int(rng.integers(0, 2))
    exact = bool(rng.integers(0, 3) == 0)
    inplace = bool(rng.integers(0, 2))
    view = int(rng.integers(0, 3))
    x = (rng.standard_normal(shape) * 10).astype(dtype) if np.dtype(dtype).kind == "f" else (rng.random(shape) * 200).astype(dtype)
    xd = torch.from_numpy(x).to(dev)
    src = xd
    xs = x
    if view == 1 and nd >= 2:                    # transposed view
        src = xd.transpose(0, nd - 1); xs = np.swapaxes(x, 0, nd - 1)
    elif view == 2:                              # every second sample along the last axis
        src = xd[..., ::2]; xs = x[..., ::2]
    if xs.shape[axis] == 0 or xs.size == 0:
        continue
    desc = "case %d: shape=%s view=%d axis=%d order=%d %s transpose=%d exact=%d inplace=%d" % (
        case, xs.shape, view, axis, order, np.dtype(dtype).name, transpose, exact, inplace)
    try:
        if transpose:
            want = np.zeros(xs.shape, dtype=dtype)
            orc.spline_filter1d_grad(np.ascontiguousarray(xs), want, axis, order)
        else:
            want = np.zeros(xs.shape, dtype=dtype)
            if order > 1:
                scipy.ndimage.spline_filter1d(np.ascontiguousarray(xs), order=order, axis=axis, output=want)
            else:
                want[...] = xs
        flag = _lib.FLAG_EXACT if exact else _lib.FLAG_AUTO
        if inplace:
            buf = src.clone() if view == 0 else src        # views: filter the view itself in place
            keep = xd.clone()
            _lib.spline_filter1d(dgm._desc(buf), dgm._desc(buf), axis, order, transpose, flag, stream)
            got = buf.cpu().numpy()
            if view == 2:                                   # the skipped samples must be untouched
                assert torch.equal(xd[..., 1::2], keep[..., 1::2]), "in-place view wrote outside the view"
        else:
            out = torch.empty(xs.shape, dtype=src.dtype, device=dev)
            _lib.spline_filter1d(dgm._desc(src), dgm._desc(out), axis, order, transpose, flag, stream)
            got = out.cpu().numpy()
        if np.dtype(dtype).kind != "f" or exact:
            np.testing.assert_array_equal(got, want)
        else:
            tol = (4e-6 if order >= 4 else 2e-6) if dtype == np.float32 else 1e-12
            err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
            scale = max(1.0, float(np.abs(want).max()))
            assert err <= tol * scale, "max abs err %.3e (scale %.3g)" % (err, scale)
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", desc)
        print("   ", str(e).strip().split("\n")[0][:300])
print("%d cases, %d failures (seed %d)" % (ncases, fails, seed))
sys.exit(1 if fails else 0)
