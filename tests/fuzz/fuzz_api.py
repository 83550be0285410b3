#!/usr/bin/env python3
"""Randomised sweep of the public API against the oracle, second part: multi-input lists with
per-input order / mode / axis, 2-D rotate / zoom, large control grids, 4 deformed axes, CUDA tensor
inputs (strided views stay on the device), crops with gradients.
python tests/fuzz/fuzz_api.py [seed] [cases]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import elasticdeform_amd as ed
from oracle import ed_oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
MODES = ["nearest", "wrap", "reflect", "mirror", "constant"]
fails = 0

def check(got, want, dtype, what, amp=1.0):
    got = got.detach().cpu().numpy() if hasattr(got, "detach") else got
    assert got.dtype == want.dtype and got.shape == want.shape, (what, got.dtype, want.dtype, got.shape, want.shape)
    if np.dtype(dtype).kind == "f":
        tol = 1e-5 if np.dtype(dtype) == np.float32 else 1e-10
        err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0
        scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
        assert err <= 2 * tol * amp * scale, "%s max abs err %.3e (scale %.3g amp %g)" % (what, err, scale, amp)
    else:
        np.testing.assert_array_equal(got, want, err_msg=what)

for case in range(ncases):
    kind = int(rng.integers(0, 5))
    try:
        if kind == 0:       # image + label list, per-input order / mode / axis, crop, affine
            nd = int(rng.choice([2, 3]))
            shape = tuple(int(rng.integers(8, {2: 120, 3: 48}[nd])) for _ in range(nd))
            img = rng.random((2,) + shape).astype(np.float32)
            lab = (rng.random(shape) * 5).astype(rng.choice([np.uint8, np.int32, np.int64]))
            disp = rng.standard_normal((nd,) + tuple(int(rng.integers(2, 5)) for _ in range(nd))) * 4
            kw = dict(order=[int(rng.integers(1, 4)), 0], mode=[str(rng.choice(MODES)), "nearest"],
                      axis=[tuple(range(1, nd + 1)), tuple(range(nd))])
            if rng.integers(0, 2):
                kw["crop"] = tuple(slice(int(n // 4), int(n // 4 + n // 2)) for n in shape)
            if rng.integers(0, 2):
                kw["affine"] = np.eye(nd, nd + 1) + rng.standard_normal((nd, nd + 1)) * 0.05
            desc = "list nd=%d shape=%s %s" % (nd, shape, kw)
            want = orc.deform_grid([img, lab], disp, **kw)
            got = ed.deform_grid([img, lab], disp, **kw)
            assert isinstance(got, list)
            check(got[0], want[0], np.float32, "img"); check(got[1], want[1], lab.dtype, "lab")
        elif kind == 1:     # 2-D rotate / zoom keywords
            shape = (int(rng.integers(10, 150)), int(rng.integers(10, 150)))
            dtype = rng.choice([np.float32, np.float64])
            X = rng.random(shape).astype(dtype)
            disp = rng.standard_normal((2, 3, 3)) * 5
            kw = dict(order=int(rng.integers(0, 6)), mode=str(rng.choice(MODES)),
                      rotate=float(rng.uniform(-60, 60)), zoom=float(rng.uniform(0.6, 1.8)))
            if rng.integers(0, 2):
                kw["crop"] = (slice(2, shape[0] - 3), slice(1, shape[1] - 2))
            desc = "rotzoom shape=%s %s %s" % (shape, np.dtype(dtype).name, kw)
            want = orc.deform_grid(X, disp, **kw)
            check(ed.deform_grid(X, disp, **kw), want, dtype, "fwd")
            dY = rng.random(want.shape).astype(dtype)
            gw = orc.deform_grid_gradient(dY, disp, X_shape=shape, **kw)
            check(ed.deform_grid_gradient(dY, disp, X_shape=shape, **kw), gw, dtype, "grad",
                  amp=64.0 if kw["order"] > 1 else 1.0)
        elif kind == 2:     # large control grids (beyond the raw-displacement / LDS-table limits)
            nd = int(rng.choice([2, 3]))
            shape = tuple(int(rng.integers(20, {2: 200, 3: 50}[nd])) for _ in range(nd))
            pts = tuple(int(rng.integers(8, {2: 70, 3: 22}[nd])) for _ in range(nd))
            dtype = rng.choice([np.float32, np.float64, np.int16])
            X = (rng.random(shape) * 50).astype(dtype)
            disp = rng.standard_normal((nd,) + pts) * 1.5
            kw = dict(order=int(rng.integers(0, 4)), mode=str(rng.choice(MODES)))
            desc = "biggrid shape=%s pts=%s %s %s" % (shape, pts, np.dtype(dtype).name, kw)
            check(ed.deform_grid(X, disp, **kw), orc.deform_grid(X, disp, **kw), dtype, "fwd")
        elif kind == 3:     # four deformed axes (exact kernels), tiny
            shape = tuple(int(rng.integers(3, 9)) for _ in range(4))
            dtype = rng.choice([np.float32, np.float64, np.uint8])
            X = (rng.random(shape) * 20).astype(dtype)
            disp = rng.standard_normal((4, 2, 2, 3, 2)) * 1.0
            kw = dict(order=int(rng.integers(0, 4)), mode=str(rng.choice(MODES)))
            desc = "4d shape=%s %s %s" % (shape, np.dtype(dtype).name, kw)
            check(ed.deform_grid(X, disp, **kw), orc.deform_grid(X, disp, **kw), dtype, "fwd")
        else:               # CUDA tensors: strided views in, tensors out, autograd
            import elasticdeform_amd.torch as et
            nd = int(rng.choice([2, 3]))
            shape = tuple(int(rng.integers(12, {2: 160, 3: 50}[nd])) for _ in range(nd))
            big = torch.from_numpy(rng.random(tuple(2 * n for n in shape)).astype(np.float32)).cuda()
            view = big[tuple(slice(1, 1 + 2 * n, 2) for n in shape)]        # stride-2 view
            Xn = view.cpu().numpy()
            disp = rng.standard_normal((nd,) + (3,) * nd) * 3
            kw = dict(order=int(rng.integers(0, 4)), mode=str(rng.choice(MODES)))
            desc = "cuda view shape=%s %s" % (shape, kw)
            y = ed.deform_grid(view, torch.from_numpy(disp).cuda(), **kw)
            assert y.is_cuda
            check(y, orc.deform_grid(Xn, disp, **kw), np.float32, "fwd")
            xa = view.clone().requires_grad_()
            ya = et.deform_grid(xa, torch.from_numpy(disp).cuda(), **kw)
            dY = torch.rand_like(ya)
            ya.backward(dY)
            gw = orc.deform_grid_gradient(dY.cpu().numpy(), disp, X_shape=shape, **kw)
            check(xa.grad, gw, np.float32, "autograd", amp=8.0 ** nd if kw["order"] > 1 else 1.0)
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", desc)
        print("   ", str(e).strip().split("\n")[0][:300])
print("%d cases, %d failures (seed %d)" % (ncases, fails, seed))
sys.exit(1 if fails else 0)
