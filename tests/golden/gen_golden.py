#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by running the REAL reference
(/root/reference's Python layer + its C extension compiled by `make -C oracle ref`) on the seeded
cases of tests/golden/cases.py.  Only runs in the build container; the fixtures it writes are
data (inputs are re-created from seeds, only reference OUTPUTS are stored).

    python tests/golden/gen_golden.py            # writes small.npz, big.npz, filters.npz, meta.json

The reference has no golden vectors of its own: its tests pin results only
relationally (vs SciPy map_coordinates, finite differences).  These fixtures pin absolute outputs,
including the legacy `nearest` / `reflect` behaviour and the integer / bool rounding rules that
nothing else on a modern SciPy can check.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases as C  # noqa: E402
from oracle import ref_loader  # noqa: E402


def main():
    ref = ref_loader.load_reference()
    ext = ref_loader.load_ref_ext()
    if ref is None or ext is None:
        print("reference not available (need /root/reference and `make -C oracle ref`); "
              "nothing generated")
        return 0
    import scipy
    import scipy.ndimage

    # GOLDEN_ONLY=small: regenerate small.npz only (the BASELINE-sized cases take minutes of CPU)
    only_small = os.environ.get("GOLDEN_ONLY") == "small"
    # GOLDEN_CASE=name: recompute that one case and merge it into the existing archive
    one_case = os.environ.get("GOLDEN_CASE")
    stores = {"small": {}, "big": {}}
    t0 = time.time()
    for case in C.all_cases():
        if only_small and case["big"]:
            continue
        if one_case and case["name"] != one_case:
            continue
        X, disp, kw = case["make"]()
        store = stores["big" if case["big"] else "small"]
        out = ref.deform_grid(X, disp, **kw)
        outs = out if isinstance(out, list) else [out]
        picks = case["pick"]() if case["pick"] else [None] * len(outs)
        for i, o in enumerate(outs):
            store["%s/out%d" % (case["name"], i)] = o if picks[i] is None else o[picks[i]].copy()
        if case["grad"]:
            dY = C.seeded_dY(case, out)
            g = ref.deform_grid_gradient(dY, disp, X_shape=C.x_shapes(X), **kw)
            gs = g if isinstance(g, list) else [g]
            gpicks = case["gpick"]() if case.get("gpick") else picks
            for i, gi in enumerate(gs):
                store["%s/grad%d" % (case["name"], i)] = \
                    gi if gpicks[i] is None else gi[gpicks[i]].copy()
        print("%-40s %6.2fs" % (case["name"], time.time() - t0), flush=True)

    # spline prefilter (SciPy, third-party arithmetic on the path) and its transpose (reference C)
    filt = {}
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 5, 8, 30, 40, 100):
        x = rng.standard_normal((3, n, 2))
        filt["x_n%d" % n] = x
        for order in range(6):
            if order > 1:
                filt["fwd_o%d_n%d" % (order, n)] = scipy.ndimage.spline_filter1d(
                    x, order=order, axis=1)
            g = np.zeros_like(x)
            ext.spline_filter1d_grad(x, g, 1, order)
            filt["tr_o%d_n%d" % (order, n)] = g
    # storage-dtype rounding of the prefilter (float32, int16, uint8 inputs; output dtype = input)
    xi = rng.random((6, 19))
    for dt in ("float32", "int16", "uint8"):
        a = (xi * 200).astype(dt) if dt != "float32" else xi.astype(dt)
        filt["xd_%s" % dt] = a
        o = np.zeros_like(a)
        scipy.ndimage.spline_filter1d(a, axis=1, order=3, output=o)
        filt["fwd_o3_%s" % dt] = o
        g = np.zeros_like(a)
        ext.spline_filter1d_grad(a, g, 1, 3)
        filt["tr_o3_%s" % dt] = g

    if one_case:
        for which in ("small", "big"):
            if stores[which]:
                path = os.path.join(HERE, which + ".npz")
                merged = dict(np.load(path))
                merged.update(stores[which])
                np.savez_compressed(path, **merged)
                with open(os.path.join(HERE, "meta.json")) as f:
                    meta = json.load(f)
                meta["n_" + which] = len(merged)
                with open(os.path.join(HERE, "meta.json"), "w") as f:
                    json.dump(meta, f, indent=1)
                print(which + ".npz", os.path.getsize(path) // 1024, "KiB,", len(merged), "arrays")
        return 0
    np.savez_compressed(os.path.join(HERE, "small.npz"), **stores["small"])
    if only_small:
        with open(os.path.join(HERE, "meta.json")) as f:
            meta = json.load(f)
        meta["n_small"] = len(stores["small"])
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
        print("small.npz", os.path.getsize(os.path.join(HERE, "small.npz")) // 1024, "KiB")
        return 0
    np.savez_compressed(os.path.join(HERE, "big.npz"), **stores["big"])
    np.savez_compressed(os.path.join(HERE, "filters.npz"), **filt)
    meta = dict(reference="gvtulder/elasticdeform v0.5.1 (/root/reference @ 2025-02-22)",
                numpy=np.__version__, scipy=scipy.__version__,
                python=sys.version.split()[0], n_small=len(stores["small"]),
                n_big=len(stores["big"]), n_filters=len(filt),
                note="outputs of the real reference; inputs re-created from seeds in cases.py")
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for n in ("small.npz", "big.npz", "filters.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)) // 1024, "KiB")
    return 0


if __name__ == "__main__":
    sys.exit(main())
