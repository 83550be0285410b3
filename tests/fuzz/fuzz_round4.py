#!/usr/bin/env python3
"""Randomised sweep over the paths that are new in round 4 -- not part of the suite; run on the GPU box:
  python tests/fuzz/fuzz_round4.py [seed] [cases]
* device-side crop window (edhip_source_window + windowed filter passes), forced to engage whatever the volume's size:
  float32 / float64 volumes with lines of 64..200 samples, random crops, all five modes, affine maps, a channel axis,
  several inputs per call -- forward against the oracle, gradient against the exact gradient (fp64 oracle);
* wide control grids (14..47 columns along x; float64 volumes from 8) on the per-strip Q tables, any shape / crop / affine map, orders 1-5;
* 16-bit float volumes that stay in 16 bits (set_reduced_precision): forward bit-equal to the float32 pipeline narrowed
  by a cast, gradient within half a 16-bit ulp of it -- and equal with the direct route switched off."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import elasticdeform_amd as ed
from oracle import ed_oracle as orc

dgm = importlib.import_module("elasticdeform_amd.deform_grid")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
MODES = ["nearest", "wrap", "reflect", "mirror", "constant"]
dev = torch.device("cuda", 0)
fails = 0
saved = (dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION)
for case in range(ncases):
    kind = ["window", "window", "half", "wide"][int(rng.integers(0, 4))]
    shape = tuple(int(rng.integers(64, 150)) for _ in range(3))
    if kind == "half":
        shape = shape[:2] + (4 * (shape[2] // 4),)
    pts = tuple(int(rng.integers(2, 6)) for _ in range(3))
    order = int(rng.choice([2, 3]))
    mode = str(rng.choice(MODES))
    sigma = float(rng.choice([0.5, 2.0, 5.0, 10.0]))
    kw = dict(order=order, mode=mode, cval=float(rng.integers(0, 3)) * 0.5)
    desc = "case %d %s shape=%s pts=%s o%d %s sigma=%g" % (case, kind, shape, pts, order, mode, sigma)
    try:
        disp = rng.standard_normal((3,) + pts) * sigma
        dd = torch.from_numpy(disp).to(dev)
        if kind == "window":
            dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION = 0.0, 1.0
            dt = np.float32 if rng.integers(0, 4) else np.float64
            crop = []
            for n in shape:
                c = int(rng.integers(4, max(5, n // 2)))
                a = int(rng.integers(0, n - c))
                crop.append(slice(a, a + c))
            kw["crop"] = tuple(crop)
            if rng.integers(0, 3) == 0:
                kw["affine"] = np.eye(3, 4) + rng.standard_normal((3, 4)) * 0.05
            full = shape
            if rng.integers(0, 3) == 0:
                full = (int(rng.integers(2, 4)),) + shape
                kw["axis"] = (1, 2, 3)
            desc += " %s crop=%s%s%s" % (np.dtype(dt).name, [(s.start, s.stop) for s in crop],
                                         " affine" if "affine" in kw else "", " ch" if "axis" in kw else "")
            X = rng.random(full).astype(dt)
            two = rng.integers(0, 3) == 0 and "axis" not in kw
            Xs = [X, (rng.random(full) * 50).astype(np.int32)] if two else X
            kws = dict(kw)
            if two:
                kws["order"] = [order, 0]
            want = orc.deform_grid(Xs, disp, **kws)
            got = ed.deform_grid([torch.from_numpy(x).to(dev) for x in Xs] if two else torch.from_numpy(X).to(dev), dd, **kws)
            w0 = want[0] if two else want
            g0 = (got[0] if two else got).cpu().numpy()
            tol = 2e-5 if dt == np.float32 else 1e-10
            err = float(np.abs(g0 - w0).max()) if w0.size else 0.0
            assert err <= tol, "forward max abs err %.3e" % err
            if two:
                assert np.array_equal(got[1].cpu().numpy(), want[1]), "label map differs"
            dY = rng.random(w0.shape).astype(dt)
            gg = ed.deform_grid_gradient(torch.from_numpy(dY).to(dev), dd, X_shape=full, **kw).cpu().numpy()
            truth = orc.deform_grid_gradient(dY.astype(np.float64), disp, X_shape=full, **kw)
            gs = max(1.0, float(np.abs(truth).max()))
            if dt == np.float32:
                gw = orc.deform_grid_gradient(dY, disp, X_shape=full, **kw)
                eref = float(np.abs(gw.astype(np.float64) - truth).max())
                egpu = float(np.abs(gg.astype(np.float64) - truth).max())
                assert egpu <= 4 * eref + 8 * np.finfo(np.float32).eps * gs, \
                    "gradient err vs exact %.3e, reference's own %.3e (scale %.3g)" % (egpu, eref, gs)
            else:
                assert float(np.abs(gg - truth).max()) <= 1e-10 * gs, "float64 gradient"
        elif kind == "wide":
            # control grids too wide for a strip's Q rows in LDS: per-strip tables (TileGeom::q_win), or the row kernel
            shape = tuple(int(rng.integers(20, 120)) for _ in range(3))
            # (float64 volumes take the per-strip tables from 8-9 columns on: wide_optional() of deform_tile.hip)
            f64 = rng.integers(0, 3) == 0
            pts = (int(rng.integers(2, 20)), int(rng.integers(2, 20)), int(rng.integers(8 if f64 else 14, 48)))
            order = int(rng.integers(1, 6))      # 1-3: per-strip tables; 4 / 5: the one-wave kernels on plain tables
            kw["order"] = order
            if rng.integers(0, 3) == 0:
                crop = []
                for n in shape:
                    a = int(rng.integers(0, max(1, n // 2)))
                    crop.append(slice(a, int(rng.integers(a + 1, n + 1))))
                kw["crop"] = tuple(crop)
            if rng.integers(0, 3) == 0:
                kw["affine"] = np.eye(3, 4) + rng.standard_normal((3, 4)) * 0.05
            kw["prefilter"] = bool(rng.integers(0, 2))
            desc = "case %d wide%s shape=%s pts=%s o%d %s sigma=%g %s" % (
                case, " float64" if f64 else "", shape, pts, order, mode, sigma,
                {k: v for k, v in kw.items() if k in ("crop", "prefilter")})
            disp = rng.standard_normal((3,) + pts) * min(sigma, 3.0)
            dd = torch.from_numpy(disp).to(dev)
            if f64:
                X = rng.random(shape)
                want = orc.deform_grid(X, disp, **kw)
                got = ed.deform_grid(torch.from_numpy(X).to(dev), dd, **kw).cpu().numpy()
                err = float(np.abs(got - want).max()) if want.size else 0.0
                assert err <= 1e-10, "float64 forward max abs err %.3e" % err
                dY = rng.random(want.shape)
                gg = ed.deform_grid_gradient(torch.from_numpy(dY).to(dev), dd, X_shape=shape, **kw).cpu().numpy()
                truth = orc.deform_grid_gradient(dY, disp, X_shape=shape, **kw)
                gs = max(1.0, float(np.abs(truth).max()))
                assert float(np.abs(gg - truth).max()) <= 1e-10 * gs, "float64 gradient"
                continue
            X = rng.random(shape).astype(np.float32)
            want = orc.deform_grid(X, disp, **kw)
            got = ed.deform_grid(torch.from_numpy(X).to(dev), dd, **kw).cpu().numpy()
            err = float(np.abs(got - want).max()) if want.size else 0.0
            assert err <= 2e-5, "forward max abs err %.3e" % err
            dY = rng.random(want.shape).astype(np.float32)
            gg = ed.deform_grid_gradient(torch.from_numpy(dY).to(dev), dd, X_shape=shape, **kw).cpu().numpy()
            gw = orc.deform_grid_gradient(dY, disp, X_shape=shape, **kw)
            truth = orc.deform_grid_gradient(dY.astype(np.float64), disp, X_shape=shape, **kw)
            gs = max(1.0, float(np.abs(truth).max()))
            eref = float(np.abs(gw.astype(np.float64) - truth).max())
            egpu = float(np.abs(gg.astype(np.float64) - truth).max())
            assert egpu <= 4 * eref + 8 * np.finfo(np.float32).eps * gs, \
                "gradient err vs exact %.3e, reference's own %.3e (scale %.3g)" % (egpu, eref, gs)
        else:
            tdt = torch.bfloat16 if rng.integers(0, 2) else torch.float16
            ulp = 2.0 ** -7 if tdt == torch.bfloat16 else 2.0 ** -10
            desc += " %s" % str(tdt).split(".")[1]
            X = torch.from_numpy(rng.random(shape).astype(np.float32)).to(dev).to(tdt)
            dY = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev).to(tdt)
            prev = ed.set_reduced_precision(True)
            try:
                got = ed.deform_grid(X, dd, **kw)
                g = ed.deform_grid_gradient(dY, dd, **kw)
                real = dgm._direct16
                dgm._direct16 = lambda *a, **k: False
                try:
                    got_cast = ed.deform_grid(X, dd, **kw)
                    g_cast = ed.deform_grid_gradient(dY, dd, **kw)
                finally:
                    dgm._direct16 = real
            finally:
                ed.set_reduced_precision(prev)
            want = ed.deform_grid(X.float(), dd, **kw).to(tdt)
            assert torch.equal(got, want), "forward: 16-bit route != float32 pipeline narrowed"
            assert torch.equal(got_cast, want), "forward: cast route != float32 pipeline narrowed"
            gw = ed.deform_grid_gradient(dY.float(), dd, **kw)
            gs = max(1.0, float(gw.abs().max()))
            for name, gx in (("16-bit route", g), ("cast route", g_cast)):
                e = float((gx.float() - gw).abs().max())
                assert e <= ulp * gs * 0.51 + 2e-5 * gs, "gradient (%s): %.3e at scale %.3g" % (name, e, gs)
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", desc)
        print("   ", str(e).strip().split("\n")[0][:300])
    finally:
        dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION = saved
# profiling build only: the tables kernel counts control columns a wide grid's per-strip window did not hold
try:
    from elasticdeform_amd import _lib
    import ctypes
    fn = _lib.load().edhip_debug_wide_clamped
    fn.restype = ctypes.c_uint
    clamped = int(fn())
    if clamped:
        fails += 1
        print("FAIL: %d control columns outside a per-strip Q window (TileGeom::q_win under-sized)" % clamped)
except AttributeError:
    pass
print("%d cases, %d failures (seed %d)" % (ncases, fails, seed))
sys.exit(1 if fails else 0)
