#!/usr/bin/env python3
"""Randomised parity sweep: elasticdeform_amd (GPU) against the oracle on random configurations --
dimensionality, shapes, control grids, orders, modes, crops, affine maps, dtypes, channel axes,
strided / transposed inputs, multi-input lists.  Not part of the test suite (minutes of oracle
time); run it on the GPU box:  python tests/fuzz/fuzz_parity.py [seed] [cases]"""
import sys, os, itertools, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import elasticdeform_amd as ed
from oracle import ed_oracle as orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
MODES = ["nearest", "wrap", "reflect", "mirror", "constant"]
fails = 0
for case in range(ncases):
    nd = int(rng.choice([1, 2, 2, 3, 3, 3]))
    big = int(os.environ.get("FUZZ_BIG", "1"))
    shape = tuple(int(rng.integers(2, {1: 400 * big, 2: 150 * big, 3: 60 * big}[nd])) for _ in range(nd))
    pts = tuple(int(rng.integers(1, 6)) for _ in range(nd))
    dtype = rng.choice([np.float32, np.float32, np.float64, np.int16, np.uint8, np.int32, np.bool_])
    order = int(rng.integers(0, 6))
    mode = str(rng.choice(MODES))
    sigma = float(rng.choice([0.5, 3.0, 8.0, 25.0]))
    disp = (rng.standard_normal((nd,) + pts) * sigma).astype(rng.choice([np.float64, np.float32]))
    kw = dict(order=order, mode=mode, cval=float(rng.integers(0, 3)), prefilter=bool(rng.integers(0, 2)))
    # optional channel axis in front or behind
    ch = int(rng.integers(0, 3))
    full = shape
    axis = None
    if ch == 1:
        full = (int(rng.integers(1, 4)),) + shape; axis = tuple(range(1, nd + 1))
    elif ch == 2:
        full = shape + (int(rng.integers(1, 4)),); axis = tuple(range(0, nd))
    if axis is not None:
        kw["axis"] = axis
    if rng.integers(0, 3) == 0:
        crop = []
        for n in shape:
            a = int(rng.integers(0, max(1, n // 2))); b = int(rng.integers(a + 1, n + 1))
            crop.append(slice(a, b))
        kw["crop"] = tuple(crop)
    if nd > 1 and rng.integers(0, 3) == 0:
        kw["affine"] = np.eye(nd, nd + 1) + rng.standard_normal((nd, nd + 1)) * 0.08
    if np.dtype(dtype) == np.bool_:
        X = rng.random(full) > 0.5
    elif np.dtype(dtype).kind == "f":
        X = rng.random(full).astype(dtype)
    else:
        X = (rng.random(full) * 200).astype(dtype)
    lay = int(rng.integers(0, 4))
    if lay == 1:
        X = np.asfortranarray(X)
    elif lay == 2 and X.ndim >= 2:
        X = np.ascontiguousarray(X.swapaxes(0, 1)).swapaxes(0, 1)
    desc = "case %d: nd=%d shape=%s pts=%s %s o%d %s sigma=%g ch=%d lay=%d %s" % (
        case, nd, full, pts, np.dtype(dtype).name, order, mode, sigma, ch, lay,
        {k: v for k, v in kw.items() if k not in ("order", "mode")})
    try:
        want = orc.deform_grid(X, disp, **kw)
        got = ed.deform_grid(X, disp, **kw)
        if np.dtype(dtype).kind == "f":
            tol = 1e-5 if dtype == np.float32 else 1e-10
            scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
            err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0
            assert err <= tol * scale * 2, "forward max abs err %.3e (scale %.3g)" % (err, scale)
            dY = rng.random(want.shape).astype(dtype)
            gw = orc.deform_grid_gradient(dY, disp, X_shape=full, **kw)
            gg = ed.deform_grid_gradient(dY, disp, X_shape=full, **kw)
            gs = max(1.0, float(np.abs(gw).max())) if gw.size else 1.0
            err = float(np.abs(gg.astype(np.float64) - gw.astype(np.float64)).max()) if gw.size else 0.0
            if dtype == np.float32 and gw.size:
                # measured bound (tests/test_gpu_parity.py:_f32_grad_check): no further from the exact
                # gradient than 4x the reference's own float32 evaluation
                truth = orc.deform_grid_gradient(dY.astype(np.float64), disp, X_shape=full, **kw)
                eref = float(np.abs(gw.astype(np.float64) - truth).max())
                egpu = float(np.abs(gg.astype(np.float64) - truth).max())
                assert egpu <= 4 * eref + 4 * np.finfo(np.float32).eps * gs, \
                    "gradient err vs exact %.3e, reference's own %.3e (scale %.3g)" % (egpu, eref, gs)
            else:
                assert err <= tol * gs * 2, "gradient max abs err %.3e (scale %.3g)" % (err, gs)
        else:
            np.testing.assert_array_equal(got, want)
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", desc)
        print("   ", str(e).strip().split("\n")[0][:300])
        tb = traceback.format_exc().strip().split("\n")
        print("   ", tb[-1][:300])
print("%d cases, %d failures (seed %d)" % (ncases, fails, seed))
sys.exit(1 if fails else 0)
