#!/usr/bin/env python3
"""CPU study for the crop-aware prefilter's source box: how tight is the convex hull of the control
coefficients, and how fast does it tighten under cubic B-spline subdivision (Lane-Riesenfeld: every level
halves the knot spacing; new points (c[i] + c[i+1]) / 2 and (c[i-1] + 6 c[i] + c[i+1]) / 8)?
Mirror-extended 5^3 grids, N(0, sigma) raw values, the crop's range of the control coordinate per axis."""
import numpy as np
import scipy.ndimage


def subdivide_axis(c, axis):
    c = np.moveaxis(c, axis, 0)
    n = c.shape[0]
    out = np.empty((2 * n - 5,) + c.shape[1:])          # points whose neighbours exist: v1 e1 v2 ... v(n-2)
    out[0::2] = (c[:-2] + 6 * c[1:-1] + c[2:]) / 8      # vertex points for i = 1 .. n-2
    out[1::2] = (c[1:-2] + c[2:-1]) / 2                 # edge points between i and i+1, i = 1 .. n-3
    return np.moveaxis(out, 0, axis)


def study(ncp=5, sigma=5.0, n=256, crop=128, seeds=range(8), levels=3):
    rows = []
    for seed in seeds:
        rng = np.random.default_rng(seed)
        raw = rng.standard_normal((ncp,) * 3) * sigma
        coef = scipy.ndimage.spline_filter(raw, order=3, mode="mirror")
        lo = (n - crop) // 2
        # exact range of the spline over the crop's voxels
        o = np.arange(lo, lo + crop, dtype=np.float64) * (ncp - 1) / (n - 1)
        grid = np.meshgrid(o, o, o, indexing="ij")
        d = scipy.ndimage.map_coordinates(coef, grid, order=3, mode="mirror", prefilter=False)
        exact = (d.min(), d.max())
        # control points that reach the crop (mirror-extended by 3 on each side to keep it simple)
        ext = np.pad(coef, 3, mode="reflect")
        a = int(np.floor(o[0])) - 1 + 3
        b = int(np.floor(o[-1])) + 2 + 3
        sub = ext[a:b + 1, a:b + 1, a:b + 1]
        hull = [(sub.min(), sub.max())]
        # subdivision keeps the curve; each level the points needed for the crop's parameter range shrink to it
        cur, t0, h = ext, -3.0, 1.0                     # cur[i] sits at parameter t0 + i * h
        for _ in range(levels):
            for ax in range(3):
                cur = subdivide_axis(cur, ax)
            t0, h = t0 + h, h / 2                       # first kept point was index 1
            ia = int(np.floor((o[0] - t0) / h)) - 1
            ib = int(np.floor((o[-1] - t0) / h)) + 2
            s = cur[ia:ib + 1, ia:ib + 1, ia:ib + 1]
            hull.append((s.min(), s.max()))
        rows.append((exact, hull))
    return rows


for ncp, sigma, crop in ((5, 5.0, 128), (5, 5.0, 64), (3, 5.0, 128), (8, 5.0, 128)):
    rows = study(ncp=ncp, sigma=sigma, crop=crop)
    ex = np.mean([e[1] - e[0] for e, _ in rows])
    line = "grid %d^3 sigma %.0f, crop %d^3 of 256^3: exact range %.1f voxels wide; hull width" % (ncp, sigma, crop, ex)
    for lvl in range(4):
        w = np.mean([h[lvl][1] - h[lvl][0] for _, h in rows])
        ok = all(h[lvl][0] <= e[0] + 1e-9 and h[lvl][1] >= e[1] - 1e-9 for e, h in rows)
        line += "  L%d %.1f%s" % (lvl, w, "" if ok else " (NOT a superset!)")
    print(line)
