"""
The reference's own test matrix (/root/reference/tests/test_deform_grid.py)
written from scratch against elasticdeform_amd -- independent of this repository's oracle:

  * forward results against pure SciPy: dense coordinates from np.linspace + map_coordinates of
    each displacement component (order 3), then map_coordinates of the image (the construction
    the reference's tests use, :36-72).  Modern SciPy changed its 'reflect' / 'nearest' spline
    handling, so -- exactly like the reference's file (:29-32,94-96) -- those two modes are left
    out of the SciPy comparisons; tests/golden/ (outputs of the real reference) pins them.
  * gradients against finite differences in float64 (:325-353): the operator is linear in X, so
    perturbing every input element and projecting on a random image is exact up to rounding.
  * the torch wrapper against the direct calls (:470-560).

Tolerances are the reference's: rtol 1e-5, atol 1e-8 (float64), atol 1e-6 where float32 is mixed in.
Everything runs on the GPU through the C ABI.
"""
import itertools
import os
import sys

import numpy as np
import pytest
import scipy.ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import elasticdeform_amd as elasticdeform  # noqa: E402

MODES_SCIPY = ["mirror", "wrap", "constant"]       # still agree with modern SciPy
RNG = np.random.default_rng(20240917)


def deform_grid_py(X, displacement, order=3, mode="constant", cval=0.0, crop=None, prefilter=True,
                   axis=None):
    """SciPy restatement of deform_grid for one array (what the reference tests compare with)."""
    if axis is None:
        axis = tuple(range(X.ndim))
    elif isinstance(axis, int):
        axis = (axis,)
    points = displacement.shape[1:]
    shape = [X.shape[a] for a in axis]
    xi = np.meshgrid(*[np.linspace(0, p - 1, s) for p, s in zip(points, shape)], indexing="ij")
    coords = list(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"))
    for i in range(len(shape)):
        coords[i] = coords[i] + scipy.ndimage.map_coordinates(displacement[i], xi, order=3)
    if crop is not None:
        coords = [c[tuple(crop)] for c in coords]
    if len(axis) == X.ndim:
        return scipy.ndimage.map_coordinates(X, coords, order=order, mode=mode, cval=cval,
                                             prefilter=prefilter)
    # loop over the non-deformed axes
    other = [a for a in range(X.ndim) if a not in axis]
    out_shape = list(X.shape)
    for a, c in zip(axis, coords[0].shape):
        out_shape[a] = c
    out = np.zeros(out_shape, dtype=X.dtype)
    for idx in itertools.product(*[range(X.shape[a]) for a in other]):
        sl = [slice(None)] * X.ndim
        for a, i in zip(other, idx):
            sl[a] = i
        out[tuple(sl)] = scipy.ndimage.map_coordinates(X[tuple(sl)], coords, order=order, mode=mode,
                                                       cval=cval, prefilter=prefilter)
    return out


def run_comparison(shape, points, order=3, sigma=25, crop=None, mode="constant", axis=None, cval=0.0,
                   dtype=np.float64, atol=1e-8):
    X = RNG.random(shape).astype(dtype)
    naxis = len(shape) if axis is None else len(axis)
    if not isinstance(points, (list, tuple)):
        points = [points] * naxis
    displacement = RNG.standard_normal((naxis,) + tuple(points)) * sigma
    got = elasticdeform.deform_grid(X, displacement, order=order, mode=mode, cval=cval, crop=crop, axis=axis)
    want = deform_grid_py(X, displacement, order=order, mode=mode, cval=cval, crop=crop, axis=axis)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=atol)


def test_random():
    for shape, points in (((100, 100), 3), ((100, 75), (3, 5))):
        X = RNG.random(shape)
        Y = elasticdeform.deform_random_grid(X, points=points)
        assert Y.shape == X.shape and Y.dtype == X.dtype


def test_basic_2d():
    for points in ((3, 3), (3, 5), (1, 5)):
        for shape in ((100, 100), (100, 75)):
            for order in range(5):
                for mode in MODES_SCIPY:
                    run_comparison(shape, points, order=order, mode=mode)


def test_basic_3d():
    for points in ((3, 3, 3), (3, 5, 7), (1, 3, 5)):
        for shape in ((50, 50, 50), (100, 50, 25)):
            for order in range(5):
                run_comparison(shape, points, order=order, mode="constant", sigma=5)


def test_crop_2d():
    for crop in ((slice(0, 50), slice(0, 50)), (slice(20, 60), slice(20, 60)), (slice(50, 100), slice(0, 100))):
        for order in range(5):
            run_comparison((100, 100), (3, 3), order=order, crop=crop)


def test_crop_3d():
    run_comparison((25, 25, 25), (3, 3, 5), order=3, crop=(slice(15, 25), slice(None), slice(None)), sigma=3)


def test_crop_rotate_zoom():
    """full[crop] == cropped for EVERY rotate / zoom / affine combination, as the reference asserts it (:121-133): the
    crop keeps the centre of the output where it was ((10 + 90) / 2 == (20 + 80) / 2 == 50), and rotate / zoom act
    about the centre of the (cropped) output (deform_grid.py:401-438), so the transform commutes with this crop."""
    crop = (slice(10, 90), slice(20, 80))
    for rotate in (-30, 0, 30, None):
        for zoom in (0.5, 1.0, 1.5, None):
            for affine in (None, np.eye(3)):
                X = RNG.random((100, 100))
                displacement = RNG.standard_normal((2, 3, 3)) * 3
                kw = dict(rotate=rotate, zoom=zoom, affine=affine)
                full = elasticdeform.deform_grid(X, displacement, **kw)
                part = elasticdeform.deform_grid(X, displacement, crop=crop, **kw)
                assert part.shape == (80, 60)
                np.testing.assert_allclose(full[crop], part, rtol=1e-5, atol=1e-8,
                                           err_msg="rotate=%r zoom=%r affine=%s" % (rotate, zoom, affine is not None))


def test_multi_2d():
    X = RNG.random((100, 75))
    Y = RNG.random((100, 75)).astype(np.float32)
    displacement = RNG.standard_normal((2, 3, 3)) * 25
    for order in list(range(5)) + [[0, 3]]:
        for crop in (None, (slice(15, 25), slice(None))):
            for cval in (0.0, 1.0, [0.0, 1.0]):
                for mode in ("constant", ["constant", "mirror"]):
                    res = elasticdeform.deform_grid([X, Y], displacement, order=order, crop=crop, cval=cval, mode=mode)
                    assert isinstance(res, list) and res[0].dtype == X.dtype and res[1].dtype == Y.dtype
                    for i, Z in enumerate((X, Y)):
                        o = order[i] if isinstance(order, list) else order
                        c = cval[i] if isinstance(cval, list) else cval
                        m = mode[i] if isinstance(mode, list) else mode
                        want = deform_grid_py(Z, displacement, order=o, cval=c, mode=m, crop=crop)
                        np.testing.assert_allclose(res[i], want, rtol=1e-5, atol=1e-8 if i == 0 else 1e-5)


def test_multi_3d():
    X = RNG.random((25, 25, 30))
    Y = RNG.random((25, 25, 30))
    displacement = RNG.standard_normal((3, 3, 3, 3)) * 3
    for order in range(5):
        for crop in (None, (slice(15, 25), slice(None), slice(None))):
            res = elasticdeform.deform_grid([X, Y], displacement, order=order, crop=crop)
            for Z, R in zip((X, Y), res):
                np.testing.assert_allclose(R, deform_grid_py(Z, displacement, order=order, crop=crop),
                                           rtol=1e-5, atol=1e-8)


def test_different_strides():
    X = RNG.random((200, 150))
    displacement = RNG.standard_normal((2, 3, 3)) * 25
    a = elasticdeform.deform_grid(np.ascontiguousarray(X), displacement, prefilter=False)
    b = elasticdeform.deform_grid(np.asfortranarray(X), displacement, prefilter=False)
    np.testing.assert_array_equal(a, b)      # the reference only runs these; layouts must agree


def test_axis():
    for shape, axis in (((30, 20, 3), (0, 1)), ((20, 3, 30), (0, 2)), ((200, 3, 100, 4), (0, 2))):
        run_comparison(shape, (3, 3), axis=axis, sigma=5)
        run_comparison(shape, (3, 3), axis=axis, sigma=5, crop=(slice(3, 15), slice(2, 18)))
    # several inputs, same / different axes
    displacement = RNG.standard_normal((2, 3, 3)) * 5
    A, B = RNG.random((3, 90, 80, 7)), RNG.random((7, 90, 80))
    res = elasticdeform.deform_grid([A, B], displacement, axis=(1, 2))
    np.testing.assert_allclose(res[0], deform_grid_py(A, displacement, axis=(1, 2)), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(res[1], deform_grid_py(B, displacement, axis=(1, 2)), rtol=1e-5, atol=1e-8)
    C, D = RNG.random((3, 20, 30)), RNG.random((20, 30))
    res = elasticdeform.deform_grid([C, D], displacement, axis=[(1, 2), (0, 1)])
    np.testing.assert_allclose(res[0], deform_grid_py(C, displacement, axis=(1, 2)), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(res[1], deform_grid_py(D, displacement, axis=(0, 1)), rtol=1e-5, atol=1e-8)


def verify_grad(shape, displacement, X_shape=None, **kw):
    """Finite differences on every input element, projected on a random image (:325-353)."""
    X = RNG.random(shape)
    Y = elasticdeform.deform_grid(X, displacement, **kw)
    R = RNG.random(Y.shape)
    grad = elasticdeform.deform_grid_gradient(R, displacement, X_shape=X.shape if X_shape else None, **kw)
    assert grad.shape == X.shape
    eps = 1e-4
    base = float((Y * R).sum())
    # the map is linear in X: one batched evaluation of all perturbations keeps this fast
    flat = np.zeros(X.size)
    Xd = torch.from_numpy(X).cuda()
    Dd = torch.from_numpy(displacement).cuda()
    Rd = torch.from_numpy(R).cuda()
    for i in range(X.size):
        Xp = Xd.clone()
        Xp.view(-1)[i] += eps
        flat[i] = (float((elasticdeform.deform_grid(Xp, Dd, **kw) * Rd).sum()) - base) / eps
    np.testing.assert_allclose(grad.reshape(-1), flat, rtol=1e-5, atol=1e-6)


def test_grad_2d():
    displacement = RNG.standard_normal((2, 3, 5)) * 3
    for order in range(5):
        for mode in ("nearest", "wrap", "reflect", "mirror", "constant"):
            verify_grad((30, 25), displacement, order=order, mode=mode)


def test_grad_crop():
    displacement = RNG.standard_normal((2, 3, 3)) * 3
    for crop in ((slice(0, 10), slice(0, 10)), (slice(5, 15), slice(3, 20)), (slice(10, 20), slice(None))):
        verify_grad((20, 20), displacement, X_shape=True, crop=crop)


def test_grad_zoom_rotate():
    displacement = RNG.standard_normal((2, 3, 5)) * 3
    for zoom in (0.5, 1, 1.5):
        verify_grad((30, 25), displacement, zoom=zoom)
    for rotate in (-20, 0, 20):
        verify_grad((30, 25), displacement, rotate=rotate)


def test_grad_with_list():
    X = RNG.random((100, 75))
    Y = RNG.random((100, 75)).astype(np.float32)
    displacement = RNG.standard_normal((2, 3, 3)) * 25
    both = elasticdeform.deform_grid_gradient([X, Y], displacement)
    np.testing.assert_allclose(both[0], elasticdeform.deform_grid_gradient(X, displacement), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(both[1], elasticdeform.deform_grid_gradient(Y, displacement), rtol=1e-5, atol=1e-4)


def test_basic_2d_torch():
    import elasticdeform_amd.torch as etorch
    for order in range(3):
        for crop in (None, (slice(20, 60), slice(10, 90))):
            for mode in ("nearest", "wrap", "reflect", "mirror", "constant"):
                X = RNG.random((100, 100))
                displacement = RNG.standard_normal((2, 3, 3)) * 25
                want = elasticdeform.deform_grid(X, displacement, order=order, crop=crop, mode=mode)
                dY = RNG.random(want.shape)
                want_g = elasticdeform.deform_grid_gradient(dY, displacement, order=order, crop=crop, mode=mode,
                                                            X_shape=X.shape)
                Xt = torch.from_numpy(X).cuda().requires_grad_()
                Yt = etorch.deform_grid(Xt, torch.from_numpy(displacement), order=order, crop=crop, mode=mode)
                Yt.backward(torch.from_numpy(dY).cuda())
                np.testing.assert_almost_equal(Yt.detach().cpu().numpy(), want)
                np.testing.assert_almost_equal(Xt.grad.cpu().numpy(), want_g)


def test_multi_2d_torch():
    import elasticdeform_amd.torch as etorch
    for order in list(range(5)) + [[0, 3]]:
        for crop in (None, (slice(15, 25), slice(None))):
            for mode in ("constant", ["constant", "mirror"]):
                X, Y = RNG.random((100, 75)), RNG.random((100, 75))
                displacement = RNG.standard_normal((2, 3, 3)) * 25
                want = elasticdeform.deform_grid([X, Y], displacement, order=order, crop=crop, mode=mode)
                res = etorch.deform_grid([torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()],
                                         torch.from_numpy(displacement), order=order, crop=crop, mode=mode)
                assert isinstance(res, tuple)
                for r, w in zip(res, want):
                    np.testing.assert_almost_equal(r.cpu().numpy(), w)
