#!/usr/bin/env python3
"""Randomised sweep of the integer fast path (8- / 16-bit volumes, orders 1-5: wave-per-tile kernel with
fp64 taps + exact re-evaluation of near-tie voxels, and the exact integer prefilter on LDS line tiles)
against the oracle, BIT for bit: shapes that are ragged w.r.t. the tiles, every mode, weak to violent
deformations, crops, affine maps, a channel axis, prefilter on and off, constants that are not
integers.  Not part of the suite; run on the GPU box:  python tests/fuzz/fuzz_int.py [seed] [cases]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import elasticdeform_amd as ed
from oracle import ed_oracle as orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rng = np.random.default_rng(seed)
MODES = ["nearest", "wrap", "reflect", "mirror", "constant"]
dev = torch.device("cuda", 0)
fails = 0
for case in range(ncases):
    shape = tuple(int(rng.integers(2, 150)) for _ in range(3))
    while np.prod(shape) > 400000:
        shape = tuple(max(2, s * 3 // 4) for s in shape)
    if rng.integers(0, 6) == 0:       # a long axis: the prefilter's 129 .. 313-sample line tiles
        k = int(rng.integers(0, 3))
        shape = tuple(int(rng.integers(130, 314)) if i == k else max(2, min(s, 40)) for i, s in enumerate(shape))
    pts = tuple(int(rng.integers(1, 7)) for _ in range(3))
    order = int(rng.integers(1, 6))
    mode = str(rng.choice(MODES))
    sigma = float(rng.choice([0.0, 0.5, 2.0, 5.0, 12.0, 30.0]))
    dtype = rng.choice([np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32])
    info = np.iinfo(dtype)
    kw = dict(order=order, mode=mode, cval=float(rng.integers(-3, 4)) * 0.75, prefilter=bool(rng.integers(0, 2)))
    full = shape
    if rng.integers(0, 4) == 0:
        full = (int(rng.integers(2, 4)),) + shape
        kw["axis"] = (1, 2, 3)
    if rng.integers(0, 3) == 0:
        crop = []
        for n in shape:
            a = int(rng.integers(0, max(1, n // 2)))
            b = int(rng.integers(a + 1, n + 1))
            crop.append(slice(a, b))
        kw["crop"] = tuple(crop)
    if rng.integers(0, 3) == 0:
        kw["affine"] = np.eye(3, 4) + rng.standard_normal((3, 4)) * 0.08
    kind = int(rng.integers(0, 3))
    if kind == 0:        # full-range noise
        X = rng.integers(info.min, info.max, full, endpoint=True).astype(dtype)
    elif kind == 1:      # a few labels with large constant regions (ties and exact integers are common)
        X = (rng.integers(0, 3, tuple((s + 7) // 8 for s in full)).astype(dtype) * (info.max // 2))
        for a in range(len(full)):
            X = np.repeat(X, 8, axis=a)
        X = X[tuple(slice(0, s) for s in full)].copy()
    else:                # a ramp
        X = (np.indices(full).sum(0) % (int(info.max) + 1)).astype(dtype)
    disp = rng.standard_normal((3,) + pts) * sigma
    if rng.integers(0, 5) == 0:
        disp = np.round(disp * 2) / 2          # half-integer shifts: coordinates ON the decision boundaries
    desc = "case %d: %s shape=%s pts=%s o%d %s sigma=%g kind=%d %s" % (
        case, np.dtype(dtype).name, full, pts, order, mode, sigma, kind,
        {k: v for k, v in kw.items() if k not in ("order", "mode")})
    try:
        want = orc.deform_grid(X, disp, **kw)
        got = ed.deform_grid(torch.from_numpy(X).to(dev), torch.from_numpy(disp).to(dev), **kw).cpu().numpy()
        assert got.dtype == want.dtype and got.shape == want.shape
        bad = int((got != want).sum())
        assert bad == 0, "%d of %d voxels differ (max |diff| %d)" % (
            bad, want.size, int(np.abs(got.astype(np.int64) - want.astype(np.int64)).max()))
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", desc)
        print("   ", str(e).strip().split("\n")[0][:300])
        print("   ", traceback.format_exc().strip().split("\n")[-1][:300])
print("%d cases, %d failures (seed %d)" % (ncases, fails, seed))
sys.exit(1 if fails else 0)
