#!/usr/bin/env python3
"""Randomised sweep aimed at the float32 hot kernels (deform_hot.hip) and the single-launch batch
path: 3-D float32 volumes large enough for the LDS-tiled kernels, orders 1-5, every mode, weak to
violent deformations (tiles overflow into the spill levels), crops, affine maps, a channel axis
(step loop), batches with one grid per sample -- against the oracle.  Not part of the suite; run on
the GPU box:  python tests/fuzz/fuzz_hot.py [seed] [cases]"""
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
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
MODES = ["nearest", "wrap", "reflect", "mirror", "constant"]
dev = torch.device("cuda", 0)
fails = 0
for case in range(ncases):
    shape = tuple(int(rng.integers(9, 90)) for _ in range(3))
    while np.prod(shape) > 250000:
        shape = tuple(max(9, s * 3 // 4) for s in shape)
    pts = tuple(int(rng.integers(1, 7)) for _ in range(3))
    order = int(rng.integers(1, 6))
    mode = str(rng.choice(MODES))
    sigma = float(rng.choice([0.5, 2.0, 5.0, 12.0, 30.0]))
    kw = dict(order=order, mode=mode, cval=float(rng.integers(0, 3)) * 0.5, prefilter=bool(rng.integers(0, 2)))
    B = int(rng.choice([1, 1, 2, 3, 5]))
    ch = int(rng.integers(0, 3)) if B == 1 else 0
    full = shape
    if ch == 1:
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
    # every other case on the z-walk forward route (EDHIP_FLAG_STRONG_FIELD: csrc/deform_k1z.hip serves the geometry
    # wherever it can), the others on the default routing
    strong = bool(rng.integers(0, 2))
    if os.environ.get("FUZZ_FIELD_STRENGTH"):          # (debugging aid: the same cases with the route forced)
        strong = os.environ["FUZZ_FIELD_STRENGTH"] == "strong"
    ed.set_field_strength("strong" if strong else "auto")
    desc = "case %d%s: B=%d shape=%s pts=%s o%d %s sigma=%g %s" % (
        case, " [strong]" if strong else "", B, full, pts, order, mode, sigma, {k: v for k, v in kw.items() if k not in ("order", "mode")})
    try:
        if B == 1:
            X = rng.random(full).astype(np.float32)
            disp = rng.standard_normal((3,) + pts) * sigma
            want = orc.deform_grid(X, disp, **kw)
            # one displacement tensor for both calls: the gradient takes the forward call's tile boxes
            # (EDHIP_FLAG_USE_BOXES); every fourth case changes the grid behind PyTorch's version
            # counter in between -- stale boxes, the gradient must still be that of the new grid
            dd = torch.from_numpy(disp).to(dev)
            got = ed.deform_grid(torch.from_numpy(X).to(dev), dd, **kw).cpu().numpy()
            err = float(np.abs(got - want).max()) if want.size else 0.0
            assert err <= 2e-5, "forward max abs err %.3e" % err
            disp_fwd = disp
            if rng.integers(0, 4) == 0:
                disp = rng.standard_normal((3,) + pts) * float(rng.choice([0.5, 5.0, 20.0]))
                dd.data.copy_(torch.from_numpy(disp))
                desc += " [stale boxes]"
            dY = rng.random(want.shape).astype(np.float32)
            gw = orc.deform_grid_gradient(dY, disp, X_shape=full, **kw)
            gg = ed.deform_grid_gradient(torch.from_numpy(dY).to(dev), dd, X_shape=full, **kw).cpu().numpy()
            truth = orc.deform_grid_gradient(dY.astype(np.float64), disp, X_shape=full, **kw)
            gs = max(1.0, float(np.abs(truth).max()))
            eref = float(np.abs(gw.astype(np.float64) - truth).max())
            egpu = float(np.abs(gg.astype(np.float64) - truth).max())
            if os.environ.get("FUZZ_DUMP") and not egpu <= 4 * eref + 4 * np.finfo(np.float32).eps * gs:
                np.savez(os.environ["FUZZ_DUMP"], X=X, disp_fwd=disp_fwd, disp=disp, dY=dY, truth=truth, gg=gg,
                         order=order, mode=mode, cval=kw["cval"], prefilter=kw["prefilter"])
            assert egpu <= 4 * eref + 4 * np.finfo(np.float32).eps * gs, \
                "gradient err vs exact %.3e, reference's own %.3e (scale %.3g)" % (egpu, eref, gs)
        else:
            X = rng.random((B,) + shape).astype(np.float32)
            D = rng.standard_normal((B, 3) + pts) * sigma
            Xd, Dd = torch.from_numpy(X).to(dev), torch.from_numpy(D).to(dev)
            got = ed.deform_grid_batch(Xd, Dd, **kw)
            b = int(rng.integers(0, B))
            want = orc.deform_grid(X[b], D[b], **kw)
            err = float(np.abs(got[b].cpu().numpy() - want).max()) if want.size else 0.0
            assert err <= 2e-5, "batch forward max abs err %.3e (sample %d)" % (err, b)
            for k in range(B):
                assert torch.equal(got[k], ed.deform_grid(Xd[k], Dd[k], **kw)), "batch != per-sample call (%d)" % k
            dY = torch.rand(got.shape, device=dev)
            gb = ed.deform_grid_gradient_batch(dY, Dd, X_shape=shape, **kw)
            # the batch gradient of one sample against the exact gradient, like a single call
            dyb = dY[b].cpu().numpy()
            gw = orc.deform_grid_gradient(dyb, D[b], X_shape=shape, **kw)
            truth = orc.deform_grid_gradient(dyb.astype(np.float64), D[b], X_shape=shape, **kw)
            gs = max(1.0, float(np.abs(truth).max()))
            eref = float(np.abs(gw.astype(np.float64) - truth).max())
            egpu = float(np.abs(gb[b].cpu().numpy().astype(np.float64) - truth).max())
            assert egpu <= 4 * eref + 4 * np.finfo(np.float32).eps * gs, \
                "batch gradient err vs exact %.3e, reference's own %.3e (scale %.3g)" % (egpu, eref, gs)
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("FAIL", desc)
        print("   ", str(e).strip().split("\n")[0][:300])
        print("   ", traceback.format_exc().strip().split("\n")[-1][:300])
print("%d cases, %d failures (seed %d)" % (ncases, fails, seed))
sys.exit(1 if fails else 0)
