"""
CPU tests of the oracle (oracle/ed_oracle.c): it must reproduce the committed golden vectors --
outputs of the real reference -- bit for bit, and, where the real reference is present (build
container: /root/reference + oracle/_ref), agree with it directly on freshly seeded inputs.
This is what "parity status: PINNED" in the oracle header rests on.
"""
import numpy as np
import pytest
import scipy.ndimage

import cases as C
from oracle import ed_oracle as orc
from oracle import ref_loader


def _aslist(v):
    return v if isinstance(v, list) else [v]


def _apply_pick(case, arrs, grad=False):
    pick = case.get("gpick") if grad and case.get("gpick") else case["pick"]
    if not pick:
        return arrs
    return [a[p] for a, p in zip(arrs, pick())]


SMALL = [c for c in C.all_cases() if not c["big"]]
BIG = [c for c in C.all_cases() if c["big"]]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: c["name"])
def test_oracle_matches_golden_small(case, golden):
    X, disp, kw = case["make"]()
    out = orc.deform_grid(X, disp, **kw)
    want = golden.outputs(case, "out")
    got = _apply_pick(case, _aslist(out))
    assert len(want) == len(got)
    for w, g in zip(want, got):
        assert g.dtype == w.dtype and g.shape == w.shape
        np.testing.assert_array_equal(g, w)
    if case["grad"]:
        dY = C.seeded_dY(case, out)
        grad = orc.deform_grid_gradient(dY, disp, X_shape=C.x_shapes(X), **kw)
        want = golden.outputs(case, "grad")
        got = _apply_pick(case, _aslist(grad), grad=True)
        assert len(want) == len(got)
        for w, g in zip(want, got):
            assert g.dtype == w.dtype
            np.testing.assert_array_equal(g, w)


@pytest.mark.slow
@pytest.mark.parametrize("case", BIG, ids=lambda c: c["name"])
def test_oracle_matches_golden_baseline_configs(case, golden):
    """BASELINE.json cfg2 (256^3 crops; forward + gradient at full size), cfg3 (128^3 fwd+grad), cfg4
    (multi-input, axis, affine)."""
    if not case.get("cpu_oracle", True):
        pytest.skip("full-size case restated on the CPU for one sigma only")
    X, disp, kw = case["make"]()
    out = orc.deform_grid(X, disp, **kw)
    for w, g in zip(golden.outputs(case, "out"), _apply_pick(case, _aslist(out))):
        np.testing.assert_array_equal(g, w)
    if case["grad"]:
        dY = C.seeded_dY(case, out)
        grad = orc.deform_grid_gradient(dY, disp, X_shape=C.x_shapes(X), **kw)
        for w, g in zip(golden.outputs(case, "grad"), _apply_pick(case, _aslist(grad), grad=True)):
            np.testing.assert_array_equal(g, w)


def test_prefilter_matches_golden_and_scipy(golden):
    f = golden.filters()
    for n in (1, 2, 3, 5, 8, 30, 40, 100):
        x = f["x_n%d" % n]
        for order in range(6):
            if order > 1:
                got = orc.spline_filter1d(x, order, 1)
                np.testing.assert_array_equal(got, f["fwd_o%d_n%d" % (order, n)])
                # SciPy is on the GPU box too: third-party arithmetic pinned against the library
                np.testing.assert_array_equal(
                    got, scipy.ndimage.spline_filter1d(x, order=order, axis=1))
            g = np.zeros_like(x)
            orc.spline_filter1d_grad(x, g, 1, order)
            np.testing.assert_array_equal(g, f["tr_o%d_n%d" % (order, n)])
    for dt in ("float32", "int16", "uint8"):
        a = f["xd_%s" % dt]
        np.testing.assert_array_equal(orc.spline_filter1d(a, 3, 1), f["fwd_o3_%s" % dt])
        g = np.zeros_like(a)
        orc.spline_filter1d_grad(a, g, 1, 3)
        np.testing.assert_array_equal(g, f["tr_o3_%s" % dt])


def test_prefilter_transpose_is_adjoint():
    """<F x, y> == <x, F^T y>: the property test_grad_* pin in the reference."""
    rng = np.random.default_rng(11)
    for order in (2, 3, 4, 5):
        for n in (2, 3, 7, 40, 64):
            x, y = rng.standard_normal((2, n))
            fx = orc.spline_filter1d(x, order, 0)
            fty = np.zeros_like(y)
            orc.spline_filter1d_grad(y, fty, 0, order)
            assert abs(np.dot(fx, y) - np.dot(x, fty)) < 1e-10 * max(1.0, abs(np.dot(fx, y)))


def test_oracle_vs_scipy_map_coordinates():
    """The reference's own forward check (tests/test_deform_grid.py:36-72,355-365): dense
    coordinates from map_coordinates(displacement, linspace) then map_coordinates(X, coords)."""
    rng = np.random.default_rng(5)
    for shape, points in (((40, 30), (3, 5)), ((12, 14, 10), (3, 3, 4))):
        for order in (0, 1, 2, 3, 4):
            for mode in ("wrap", "mirror", "constant"):
                X = rng.random(shape)
                disp = rng.standard_normal((len(shape),) + points) * 5
                coords = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
                xi = np.meshgrid(*[np.linspace(0, p - 1, s) for s, p in zip(shape, points)],
                                 indexing="ij")
                coords = [c + scipy.ndimage.map_coordinates(disp[i], xi, order=3)
                          for i, c in enumerate(coords)]
                # reflect / nearest changed in SciPy 1.6 and are skipped upstream too
                # (test_deform_grid.py:29-32,94-96); the golden vectors pin those instead
                want = scipy.ndimage.map_coordinates(X, coords, order=order, mode=mode)
                got = orc.deform_grid(X, disp, order=order, mode=mode)
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-8)


def test_oracle_gradient_is_adjoint():
    rng = np.random.default_rng(6)
    for order in (0, 1, 3):
        for mode in C.MODES:
            X = rng.random((12, 15))
            disp = rng.standard_normal((2, 3, 3)) * 3
            Y = orc.deform_grid(X, disp, order=order, mode=mode, cval=0.0)
            dY = rng.random(Y.shape)
            dX = orc.deform_grid_gradient(dY, disp, order=order, mode=mode, cval=0.0)
            assert abs(np.sum(Y * dY) - np.sum(dX * X)) < 1e-9


@pytest.mark.skipif(ref_loader.load_reference() is None,
                    reason="real reference only exists in the build container")
def test_oracle_vs_live_reference():
    """Fresh seeds, real reference imported from /root/reference: bit equality."""
    ref = ref_loader.load_reference()
    rng = np.random.default_rng(123)
    n = 0
    for shape, points in (((23, 31), (3, 3)), ((9, 11, 10), (2, 4, 3))):
        for order in range(6):
            for mode in C.MODES:
                for dt in (np.float64, np.float32, np.int16, np.uint8):
                    X = (rng.random(shape) * 100).astype(dt)
                    disp = rng.standard_normal((len(shape),) + points) * 3
                    crop = tuple(slice(2, s - 3) for s in shape)
                    A = np.eye(len(shape), len(shape) + 1) + rng.standard_normal(
                        (len(shape), len(shape) + 1)) * 0.05
                    kw = dict(order=order, mode=mode, cval=0.5, crop=crop, affine=A)
                    a = ref.deform_grid(X, disp, **kw)
                    b = orc.deform_grid(X, disp, **kw)
                    np.testing.assert_array_equal(a, b)
                    if np.dtype(dt).kind == "f":
                        dY = rng.random(a.shape).astype(dt)
                        ga = ref.deform_grid_gradient(dY, disp, X_shape=X.shape, **kw)
                        gb = orc.deform_grid_gradient(dY, disp, X_shape=X.shape, **kw)
                        np.testing.assert_array_equal(ga, gb)
                    n += 1
    assert n == 240
