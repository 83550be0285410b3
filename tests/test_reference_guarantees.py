"""The reference's guarantees that the default (fast) routes trade away, given back as options:

* ``set_crop_identity(True)``  -- README.md:113 ``full[crop] == cropped`` bit for bit, also where the crop-aware prefilter
  window would engage (BASELINE cfg4: 3 x 256^3 cropped to 64^3);
* ``set_gradient_accumulation('float')`` -- deform.c:953-995 adds every tap in the array's own floating-point type, so a
  contribution's precision does not depend on its neighbours (the fixed-point tile cells resolve a contribution to
  ~1.4e-10 of its tile's sum of |dY|).

Both are checked against the committed golden vectors (outputs of the real reference, tests/golden/gen_golden.py) and
against the oracle.  GPU tests: they call the product through the C ABI.
"""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases as C  # noqa: E402
from oracle import ed_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import elasticdeform_amd as ed  # noqa: E402

F32_TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def golden_big():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big.npz"))


def _case(name):
    return [c for c in C.all_cases() if c["name"] == name][0]


def test_strict_crop_identity_on_cfg4(golden_big):
    """BASELINE cfg4 (3 x 256^3 float32 image + 256^3 int32 labels, axis, crop 64^3, affine): by default the float
    image is prefiltered inside a window around the crop; with set_crop_identity(True) the cropped call returns the
    bits of the same region of the uncropped call (README.md:113), and both agree with the reference's golden output."""
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    case = _case("cfg4_multi")
    X, disp, kw = case["make"]()
    crop = kw["crop"]
    engaged = []
    orig = dgm._crop_windows

    def spy(*a, **k):
        w = orig(*a, **k)
        engaged.append(any(x is not None for x in w))
        return w
    dgm._crop_windows = spy
    # the identity itself needs a transform that commutes with the crop (rotate / zoom / affine act about the centre of
    # the CROPPED output, deform_grid.py:401-438): cfg4's inputs, displacement and crop without its affine map
    kw_id = {k: v for k, v in kw.items() if k != "affine"}
    kw_full = {k: v for k, v in kw_id.items() if k != "crop"}
    try:
        loose = ed.deform_grid(X, disp, **kw)
        assert engaged and engaged[-1], "the crop window is expected to engage on cfg4 by default"
        loose_id = ed.deform_grid(X, disp, **kw_id)
        assert engaged[-1]
        prev = ed.set_crop_identity(True)
        try:
            strict = ed.deform_grid(X, disp, **kw)
            assert not engaged[-1]
            strict_id = ed.deform_grid(X, disp, **kw_id)
            full = ed.deform_grid(X, disp, **kw_full)
        finally:
            ed.set_crop_identity(prev)
    finally:
        dgm._crop_windows = orig
    idx_img = (slice(None),) + tuple(crop)          # the image has a leading channel axis (axis=(1, 2, 3))
    np.testing.assert_array_equal(strict_id[0], full[0][idx_img])
    np.testing.assert_array_equal(strict_id[1], full[1][tuple(crop)])
    # the window changes float results by far less than the tolerance, and nothing for the label map
    np.testing.assert_allclose(loose_id[0], strict_id[0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(loose[0], strict[0], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(loose[1], strict[1])
    # golden: the reference's own cropped output of cfg4 as BASELINE.json states it (with the affine map)
    picks = case["pick"]()
    np.testing.assert_allclose(strict[0][picks[0]], golden_big["cfg4_multi/out0"], **F32_TOL)
    np.testing.assert_array_equal(strict[1][picks[1]], golden_big["cfg4_multi/out1"])


def test_strict_crop_identity_gradient_small():
    """The gradient side of the same switch on a volume small enough for the fp64 oracle: with the window forced to
    engage the default route is within tolerance, the strict route equals the whole-volume route bit for bit."""
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    rng = np.random.default_rng(5)
    shape, pts = (150, 160, 170), (3, 3, 3)
    crop = (slice(60, 84), slice(70, 90), slice(80, 110))
    X = rng.random(shape).astype(np.float32)
    disp = rng.standard_normal((3,) + pts) * 0.5
    kw = dict(order=3, mode="constant", cval=0.25)
    saving, fraction = dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION
    dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION = 0.0, 1.0
    try:
        prev = ed.set_crop_identity(True)
        try:
            part = ed.deform_grid(X, disp, crop=crop, **kw)
            full = ed.deform_grid(X, disp, **kw)
        finally:
            ed.set_crop_identity(prev)
        loose = ed.deform_grid(X, disp, crop=crop, **kw)
    finally:
        dgm.CROP_WINDOW_MIN_SAVING, dgm.CROP_WINDOW_MAX_FRACTION = saving, fraction
    np.testing.assert_array_equal(part, full[crop])
    np.testing.assert_allclose(loose, part, rtol=0, atol=1e-6)
    np.testing.assert_allclose(part, orc.deform_grid(X, disp, crop=crop, **kw), **F32_TOL)


def test_float_gradient_accumulation_keeps_relative_precision(golden_big):
    """One dY of 1e6 among values of 1e-3: in the default mode the small voxels of the spike's tile lose their
    relative precision (absolute bound, tests/test_gpu_parity.py); with set_gradient_accumulation('float') every cell
    keeps the local relative precision of the reference -- and the headline gradient golden (cfg2, 256^3) still holds."""
    rng = np.random.default_rng(77)
    shape, pts = (48, 48, 64), (3, 3, 3)
    disp = rng.standard_normal((3,) + pts) * 1.5
    dY = (rng.random(shape).astype(np.float32) + 0.5) * 1e-3
    dY[20, 20, 20] = 1e6
    kw = dict(order=3, mode="mirror", prefilter=False)
    truth = orc.deform_grid_gradient(dY.astype(np.float64), disp, **kw)
    want = orc.deform_grid_gradient(dY, disp, **kw)
    dYd, dd = torch.from_numpy(dY).cuda(), torch.from_numpy(disp).cuda()
    fixed = ed.deform_grid_gradient(dYd, dd, **kw).cpu().numpy()
    prev = ed.set_gradient_accumulation("float")
    try:
        flt = ed.deform_grid_gradient(dYd, dd, **kw).cpu().numpy()
    finally:
        ed.set_gradient_accumulation(prev)
    # cells the spike does not feed: the spike's taps reach sources within displacement + window of (20, 20, 20)
    reach = int(np.ceil(np.abs(disp).max() * 1.5)) + 4
    small = np.ones(shape, bool)
    small[max(0, 20 - reach):21 + reach, max(0, 20 - reach):21 + reach, max(0, 20 - reach):21 + reach] = False
    small &= np.abs(truth) > 1e-5
    assert small.mean() > 0.3
    rel = lambda a: float((np.abs(a[small] - truth[small]) / np.abs(truth[small])).max())      # noqa: E731
    ref_rel = rel(want)
    assert rel(flt) <= max(1e-5, 4 * ref_rel), (rel(flt), ref_rel)
    # (and the default mode really is the one that loses it inside the spike's tile: otherwise this test proves nothing)
    tile = np.zeros(shape, bool)
    tile[max(0, 16 - reach):24 + reach, max(0, 16 - reach):24 + reach, max(0, 16 - reach):32 + reach] = True
    in_tile = small & tile
    if in_tile.any():
        worst_fixed = float((np.abs(fixed[in_tile] - truth[in_tile]) / np.abs(truth[in_tile])).max())
        worst_float = float((np.abs(flt[in_tile] - truth[in_tile]) / np.abs(truth[in_tile])).max())
        assert worst_float <= max(1e-5, 4 * ref_rel) and worst_fixed > worst_float
    # golden: cfg2's gradient (the reference's own 256^3 result) in float mode
    case = _case("cfg2_grad_s5")
    X, disp2, kw2 = case["make"]()
    out = ed.deform_grid(X, disp2, **kw2)
    dY2 = C.seeded_dY(case, out)
    prev = ed.set_gradient_accumulation("float")
    try:
        g = ed.deform_grid_gradient(dY2, disp2, X_shape=C.x_shapes(X), **kw2)
    finally:
        ed.set_gradient_accumulation(prev)
    w = golden_big["cfg2_grad_s5/grad0"]
    gp = g[case["gpick"]()[0]] if "gpick" in case else g[case["pick"]()[0]]
    np.testing.assert_allclose(gp, w, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(w).max())))


def test_option_setters_validate_and_return_previous():
    assert ed.set_crop_identity(True) is False
    assert ed.set_crop_identity(False) is True
    assert ed.set_gradient_accumulation("float") == "fixed"
    assert ed.set_gradient_accumulation("fixed") == "float"
    with pytest.raises(ValueError):
        ed.set_gradient_accumulation("double")
