"""
Seeded case definitions shared by the golden-vector generator (gen_golden.py, runs the REAL
reference in the build container) and by the parity tests (which re-create the same inputs from
the same seeds and compare against the stored reference outputs).

A case is a dict:
    name     unique key
    make     () -> (X, displacement, kwargs)      X: ndarray or list of ndarrays
    grad     bool: also store deform_grid_gradient(dY, ...) for a seeded dY
    pick     optional () -> tuple-of-index-expressions applied to every stored output (to keep
             fixtures of the BASELINE-sized cases small); one entry per output
    big      True for cases that need seconds of CPU in the oracle (BASELINE.json cfg2-cfg4 shapes)

Nothing here reads /root/reference.
"""
import zlib

import numpy as np

MODES = ("nearest", "wrap", "reflect", "mirror", "constant")


def _seed(name):
    return zlib.crc32(name.encode())


def _data(rng, shape, dtype):
    dtype = np.dtype(dtype)
    if dtype == np.bool_:
        return rng.random(shape) > 0.5
    if dtype.kind == "f":
        return rng.random(shape).astype(dtype)
    if dtype.kind == "u":
        return (rng.random(shape) * 250).astype(dtype)
    return (rng.random(shape) * 400 - 150).astype(dtype)


def _affine(rng, nd):
    return np.eye(nd, nd + 1) + rng.standard_normal((nd, nd + 1)) * 0.05


def _simple(name, shape, points, dtype, order, mode, crop, affine, sigma=3.0, grad=True,
            cval=0.5):
    def make():
        rng = np.random.default_rng(_seed(name))
        X = _data(rng, shape, dtype)
        disp = rng.standard_normal((len(shape),) + tuple(points)) * sigma
        kw = dict(order=order, mode=mode, cval=cval)
        if crop is not None:
            kw["crop"] = crop
        if affine:
            kw["affine"] = _affine(rng, len(shape))
        return X, disp, kw
    return dict(name=name, make=make, grad=grad and np.dtype(dtype).kind == "f", pick=None,
                big=False)


def small_cases():
    cases = []
    # A: 2-D, every order x every mode x crop x affine, f64; f32 without affine
    shape, pts = (13, 17), (3, 4)
    crop2 = (slice(2, 10), slice(3, 15))
    for order in range(6):
        for mode in MODES:
            for ci, crop in enumerate((None, crop2)):
                for aff in (0, 1):
                    cases.append(_simple("A2d_o%d_%s_c%d_a%d_f64" % (order, mode, ci, aff),
                                         shape, pts, np.float64, order, mode, crop, aff))
                cases.append(_simple("A2d_o%d_%s_c%d_a0_f32" % (order, mode, ci),
                                     shape, pts, np.float32, order, mode, crop, 0))
    # A': a grid axis of length 1 (test_basic_2d uses points (1, 5)) and strong folding (sigma 25)
    for order in (0, 1, 3):
        for mode in ("mirror", "constant", "wrap"):
            cases.append(_simple("A2d_p15_o%d_%s" % (order, mode), (13, 17), (1, 5), np.float64,
                                 order, mode, None, 0, sigma=3.0))
            cases.append(_simple("A2d_s25_o%d_%s" % (order, mode), (13, 17), (3, 3), np.float64,
                                 order, mode, None, 0, sigma=25.0))
    # B: 3-D
    shape, pts = (7, 8, 9), (3, 3, 4)
    crop3 = (slice(1, 6), slice(0, 8), slice(2, 7))
    for order in (0, 1, 2, 3, 5):
        for mode in MODES:
            for ci, crop in enumerate((None, crop3)):
                cases.append(_simple("B3d_o%d_%s_c%d_a0_f64" % (order, mode, ci), shape, pts,
                                     np.float64, order, mode, crop, 0))
    for mode in MODES:
        cases.append(_simple("B3d_o3_%s_c1_a1_f64" % mode, shape, pts, np.float64, 3, mode,
                             crop3, 1))
        cases.append(_simple("B3d_o3_%s_c0_a0_f32" % mode, shape, pts, np.float32, 3, mode,
                             None, 0))
    cases.append(_simple("B3d_o4_mirror_c0_a0_f64", shape, pts, np.float64, 4, "mirror", None, 0))
    # 1-D and 4-D corner cases
    for order in (0, 3):
        cases.append(_simple("L1d_o%d_mirror" % order, (40,), (4,), np.float64, order, "mirror",
                             None, 0))
        cases.append(_simple("H4d_o%d_constant" % order, (5, 4, 6, 5), (2, 3, 2, 3), np.float64,
                             order, "constant", None, 0, sigma=1.0))
    # 5 to 7 deformed axes (the reference takes any number, _deform_grid.c:158-175; 8-dimensional
    # arrays allow 7 here)
    for order in (1, 3):
        cases.append(_simple("H5d_o%d_mirror" % order, (4, 3, 4, 3, 5), (2, 2, 3, 2, 2), np.float64,
                             order, "mirror", None, 0, sigma=1.0))
    cases.append(_simple("H5d_o3_reflect_f32", (4, 3, 4, 3, 5), (2, 2, 3, 2, 2), np.float32, 3,
                         "reflect", None, 0, sigma=1.0))
    cases.append(_simple("H6d_o2_constant", (3, 3, 3, 2, 3, 4), (2, 2, 2, 2, 2, 2), np.float64, 2,
                         "constant", None, 0, sigma=0.7))
    cases.append(_simple("H7d_o1_nearest", (2, 3, 2, 2, 3, 2, 3), (2, 2, 2, 2, 2, 2, 2), np.float64, 1,
                         "nearest", None, 0, sigma=0.7))
    cases.append(_simple("H5d_o0_int16", (4, 3, 4, 3, 5), (2, 2, 3, 2, 2), "int16", 0, "nearest", None,
                         0, sigma=1.0, grad=False))
    # C: integer / bool dtypes (rounding, clamping, prefilter-in-storage-dtype behaviour)
    for dt in ("int16", "uint8", "bool", "int32", "uint16", "int64", "int8", "uint32", "uint64"):
        for order in (0, 1, 3):
            for mode in ("nearest", "constant"):
                cases.append(_simple("C2d_%s_o%d_%s" % (dt, order, mode), (13, 17), (3, 4), dt,
                                     order, mode, None, 0, grad=False, cval=3.0))
    return cases


def _multi_axis_case():
    name = "D_multi_axes"

    def make():
        rng = np.random.default_rng(_seed(name))
        X = rng.random((3, 13, 17)).astype(np.float32)
        Y = (rng.random((13, 17)) * 5).astype(np.int32)
        disp = rng.standard_normal((2, 3, 4)) * 3.0
        kw = dict(order=[3, 0], mode=["mirror", "nearest"], cval=[0.0, 1.0],
                  axis=[(1, 2), (0, 1)], crop=(slice(2, 10), slice(3, 15)))
        return [X, Y], disp, kw
    return dict(name=name, make=make, grad=False, pick=None, big=False)


def _channels_last_case():
    name = "D_channels_last"

    def make():
        rng = np.random.default_rng(_seed(name))
        X = rng.random((11, 3, 12, 2))
        disp = rng.standard_normal((2, 3, 3)) * 2.0
        kw = dict(order=3, mode="reflect", axis=(0, 2))
        return X, disp, kw
    return dict(name=name, make=make, grad=True, pick=None, big=False)


def _multi_grad_case():
    name = "D_multi_grad"

    def make():
        rng = np.random.default_rng(_seed(name))
        X = rng.random((13, 17))
        Y = rng.random((13, 17)).astype(np.float32)
        disp = rng.standard_normal((2, 3, 3)) * 4.0
        kw = dict(order=[2, 3], mode=["constant", "reflect"], cval=[0.0, 1.0],
                  crop=(slice(1, 9), slice(2, 12)))
        return [X, Y], disp, kw
    return dict(name=name, make=make, grad=True, pick=None, big=False)


def _fortran_case():
    name = "D_fortran_order"

    def make():
        rng = np.random.default_rng(_seed(name))
        X = np.asfortranarray(rng.random((14, 11)))
        disp = rng.standard_normal((2, 3, 3)) * 2.0
        return X, disp, dict(order=3, mode="mirror", prefilter=False)
    return dict(name=name, make=make, grad=True, pick=None, big=False)


def _rotate_zoom_cases():
    out = []
    for rotate in (-30, 30, None):
        for zoom in (0.5, 1.5, None):
            for ci, crop in enumerate((None, (slice(3, 19), slice(5, 21)))):
                for ai in (0, 1):
                    if rotate is None and zoom is None and ai == 0:
                        continue
                    name = "E_rot%s_zoom%s_c%d_a%d" % (rotate, zoom, ci, ai)

                    def make(name=name, rotate=rotate, zoom=zoom, crop=crop, ai=ai):
                        rng = np.random.default_rng(_seed(name))
                        X = rng.random((22, 26))
                        disp = rng.standard_normal((2, 3, 3)) * 2.0
                        kw = dict(order=3, mode="constant", rotate=rotate, zoom=zoom)
                        if crop is not None:
                            kw["crop"] = crop
                        if ai:
                            kw["affine"] = np.eye(3)
                        return X, disp, kw
                    out.append(dict(name=name, make=make, grad=True, pick=None, big=False))
    return out


# ---- BASELINE.json / SURVEY.md section 8(d) configurations ---------------------------------------

def cfg1_inputs():
    X = np.zeros((200, 300), dtype=np.float32)
    X[::10, ::10] = 1
    disp = np.random.default_rng(1).standard_normal((2, 3, 3)) * 25
    return X, disp, dict(order=3, mode="constant")


def cfg2_inputs(sigma=5.0, n=256):
    X = np.random.default_rng(2).random((n, n, n), dtype=np.float32)
    disp = np.random.default_rng(22).standard_normal((3, 5, 5, 5)) * sigma
    return X, disp, dict(order=3, mode="mirror")


def cfg3_inputs(n=128):
    X = np.random.default_rng(3).random((n, n, n), dtype=np.float32)
    disp = np.random.default_rng(33).standard_normal((3, 5, 5, 5)) * 2.5
    dY = np.random.default_rng(333).random((n, n, n), dtype=np.float32)
    return X, disp, dict(order=3, mode="mirror"), dY


def cfg4_affine():
    """3x4 affine: rotation by 10 degrees about axis 2 composed with zoom 1.1, about the centre
    of the 64^3 crop (crop-local coordinates, like the reference's rotate/zoom)."""
    th = np.radians(10.0)
    R = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    M = 1.1 * R
    c = np.array([31.5, 31.5, 31.5])
    A = np.zeros((3, 4))
    A[:, :3] = M
    A[:, 3] = c - M @ c
    return A


def cfg4_inputs(n=256):
    img = np.random.default_rng(4).random((3, n, n, n), dtype=np.float32)
    lab = np.random.default_rng(44).integers(0, 4, (n, n, n)).astype(np.int32)
    disp = np.random.default_rng(45).standard_normal((3, 5, 5, 5)) * 5.0
    lo, hi = (n * 3) // 8, (n * 5) // 8
    kw = dict(order=[3, 0], mode=["mirror", "nearest"], axis=[(1, 2, 3), (0, 1, 2)],
              crop=(slice(lo, hi),) * 3, affine=cfg4_affine())
    return [img, lab], disp, kw


CFG5_BATCH = 64            # volumes per GPU of BASELINE cfg5 (512 volumes of 128^3 over 8 GPUs)
CFG5_GOLDEN = (0, 31, 63)  # samples of the shard whose reference outputs are stored


def cfg5_sample(b, n=128):
    """Sample b of a cfg5 shard: a 128^3 float32 volume, its own 5^3 control grid (sigma 2.5, the
    relative strength of cfg2 at 256^3) and a gradient seed -- one RNG stream per sample so that a
    rank can materialise just its shard (generated per shard)."""
    X = np.random.default_rng(5000 + b).random((n, n, n), dtype=np.float32)
    disp = np.random.default_rng(55000 + b).standard_normal((3, 5, 5, 5)) * 2.5
    dY = np.random.default_rng(555000 + b).random((n, n, n), dtype=np.float32)
    return X, disp, dict(order=3, mode="mirror"), dY


def big_cases():
    cases = []
    cases.append(dict(name="cfg1_readme", make=cfg1_inputs, grad=False, big=False,
                      pick=lambda: ((slice(None, None, 3), slice(None, None, 3)),)))
    for tag, crop in (("corner", (slice(0, 24),) * 3), ("inner", (slice(100, 124),) * 3),
                      ("edge", (slice(232, 256), slice(116, 140), slice(0, 24)))):
        for sigma in (5.0, 10.0):
            def make(crop=crop, sigma=sigma):
                X, disp, kw = cfg2_inputs(sigma)
                kw["crop"] = crop
                return X, disp, kw
            cases.append(dict(name="cfg2_%s_s%g" % (tag, sigma), make=make, grad=False,
                              pick=None, big=True))

    # the benchmark's own geometry, forward AND gradient at 256^3 (the headline kernel's
    # gradient had no element-wise reference at its size): outputs / gradients stored on a sub-grid
    for sigma in (5.0, 10.0):
        def make2(sigma=sigma):
            return cfg2_inputs(sigma)
        # (cpu_oracle: tests/test_oracle.py restates one of the two at full size -- 80 s of CPU each)
        cases.append(dict(name="cfg2_grad_s%g" % sigma, make=make2, grad=True, big=True, cpu_oracle=sigma == 5.0,
                          dY=lambda: np.random.default_rng(10).random((256, 256, 256), dtype=np.float32),
                          pick=lambda: ((slice(3, None, 8), slice(5, None, 8), slice(None, None, 4)),),
                          gpick=lambda: ((slice(1, None, 8), slice(6, None, 8), slice(2, None, 4)),)))

    def make3():
        X, disp, kw, _ = cfg3_inputs()
        return X, disp, kw
    cases.append(dict(name="cfg3_128", make=make3, grad=True, big=True,
                      dY=lambda: cfg3_inputs()[3],
                      pick=lambda: ((slice(40, 72, 2), slice(0, 128, 8), slice(96, 128)),)))
    for b in CFG5_GOLDEN:
        def make5(b=b):
            X, disp, kw, _ = cfg5_sample(b)
            return X, disp, kw
        cases.append(dict(name="cfg5_b%d" % b, make=make5, grad=True, big=True,
                          dY=lambda b=b: cfg5_sample(b)[3],
                          pick=lambda: ((slice(3, 128, 8), slice(5, 128, 8), slice(64, 128)),)))
    # gradient at size too: dX has the inputs' shapes (3 x 256^3 float32 accumulated
    # through the transposed prefilter, 256^3 int32), stored on a sub-grid of the region the crop maps from
    cases.append(dict(name="cfg4_multi", make=cfg4_inputs, grad=True, big=True,
                      pick=lambda: ((slice(None), slice(None, None, 4), slice(None, None, 4),
                                     slice(None, None, 2)),
                                    (slice(None, None, 2), slice(None, None, 2), slice(None))),
                      gpick=lambda: ((slice(None), slice(72, 184, 4), slice(72, 184, 4), slice(72, 184, 2)),
                                     (slice(72, 184, 2), slice(72, 184, 2), slice(72, 184)))))
    return cases


def all_cases():
    return (small_cases() + [_multi_axis_case(), _channels_last_case(), _multi_grad_case(),
                             _fortran_case()] + _rotate_zoom_cases() + big_cases())


def seeded_dY(case, outputs):
    """Gradient seed for a case: explicit for the big ones, else derived from the case name."""
    if "dY" in case:
        dY = case["dY"]()
        return dY
    rng = np.random.default_rng(_seed(case["name"] + "/dY"))
    outs = outputs if isinstance(outputs, list) else [outputs]
    dys = [rng.random(o.shape).astype(o.dtype) for o in outs]
    return dys if isinstance(outputs, list) else dys[0]


def x_shapes(X):
    return [x.shape for x in X] if isinstance(X, list) else X.shape
