"""
CPU tests of the host layer (elasticdeform_amd/_host.py): what the reference's Python helpers hand
to the C entry point (deform_grid.py:295-454) and how they fail.  No GPU, no library call.
Expected values come from the oracle's independent restatement (oracle/ed_oracle.py) and, in the
build container, from the real reference.
"""
import numpy as np
import pytest

from elasticdeform_amd import _host
from oracle import ed_oracle as orc
from oracle import ref_loader


def _plan(Xs, disp, order=3, mode="constant", cval=0.0, crop=None, axis=None, affine=None,
          rotate=None, zoom=None):
    return _host.Plan(Xs, disp, order, mode, cval, crop, axis, affine, rotate, zoom)


def test_plan_matches_oracle_restatement():
    rng = np.random.default_rng(0)
    X = rng.random((3, 22, 26))
    Y = rng.random((22, 26))
    disp = rng.standard_normal((2, 3, 3))
    crop = (slice(3, 19), slice(None, 21))
    for affine in (None, np.eye(3), np.eye(2, 3) + rng.standard_normal((2, 3)) * 0.1):
        for rotate in (None, -30, 0, 25.5):
            for zoom in (None, 0.5, 1.0, 1.5):
                p = _plan([X, Y], disp, [3, 0], ["mirror", "nearest"], [0.0, 2.0], crop,
                          [(1, 2), (0, 1)], affine, rotate, zoom)
                axis, out_shapes, offset, orders, modes, cvals, inv = orc.prepare(
                    [X, Y], disp, [3, 0], ["mirror", "nearest"], [0.0, 2.0], crop,
                    [(1, 2), (0, 1)], affine, rotate, zoom)
                assert p.axis == axis and [tuple(s) for s in p.output_shapes] == out_shapes
                np.testing.assert_array_equal(p.output_offset, offset)
                assert p.output_offset.dtype == np.int64
                np.testing.assert_array_equal(p.order, orders)
                np.testing.assert_array_equal(p.mode, modes)
                np.testing.assert_array_equal(p.cval, cvals)
                if inv is None:
                    assert p.inverse_affine is None
                else:
                    # bit-equal: the factors are multiplied in the reference's order
                    np.testing.assert_array_equal(p.inverse_affine, inv)
                    assert p.inverse_affine.shape == (2, 3)


def test_crop_offset_is_none_without_positive_start():
    X = np.zeros((10, 12))
    disp = np.zeros((2, 3, 3))
    p = _plan([X], disp, crop=(slice(0, 5), slice(None, 7)))
    assert p.output_offset is None and p.output_shapes == [[5, 7]]
    p = _plan([X], disp, crop=(slice(0, 5), slice(2, 7)))
    np.testing.assert_array_equal(p.output_offset, [0, 2])


def test_axis_forms():
    X = np.zeros((4, 5, 6))
    disp1 = np.zeros((1, 3))
    assert _plan([X], disp1, axis=1).axis == [(1,)]
    assert _plan([X], np.zeros((2, 3, 3)), axis=(0, 2)).axis == [(0, 2)]
    assert _plan([X, X], np.zeros((3, 2, 2, 2))).axis == [(0, 1, 2), (0, 1, 2)]


def test_failure_behaviour_matches_the_reference():
    X = np.zeros((10, 12))
    disp = np.zeros((2, 3, 3))
    with pytest.raises(Exception, match="numpy.ndarray or a list"):
        _host.normalize_inputs((X,))                      # tuple is not accepted (deform_grid.py:301)
    with pytest.raises(AssertionError):
        _host.normalize_inputs([])
    with pytest.raises(AssertionError):
        _plan([X], disp, order=6)
    with pytest.raises(AssertionError):
        _plan([X], disp, order=[3, 3])
    with pytest.raises(RuntimeError, match="boundary mode not supported"):
        _plan([X], disp, mode="periodic")
    with pytest.raises(AssertionError):
        _plan([X], np.zeros((3, 3, 3)))                   # first dim must equal naxis
    with pytest.raises(AssertionError):
        _plan([X], np.zeros((2, 3)))                      # ndim must be naxis + 1
    with pytest.raises(AssertionError):
        _plan([X], disp, axis=(1, 0))                     # sorted and unique
    with pytest.raises(AssertionError):
        _plan([X], disp, axis=(0, 2))                     # out of range
    with pytest.raises(AssertionError):
        _plan([X, np.zeros((10, 13))], disp)              # equal deformed shapes
    with pytest.raises(Exception, match="Crop must be a slice"):
        _plan([X], disp, crop=(slice(0, 5), 3))
    with pytest.raises(AssertionError):
        _plan([X], disp, crop=(slice(0, 5, 2), slice(None)))
    with pytest.raises(AssertionError):
        _plan([X], disp, crop=(slice(0, 11), slice(None)))
    with pytest.raises(AssertionError):
        _plan([X], disp, affine=np.eye(4))
    with pytest.raises(AssertionError, match="only implemented for 2D"):
        _plan([np.zeros((4, 5, 6))], np.zeros((3, 2, 2, 2)), rotate=10)
    # a homogeneous 4x4 in 3-D trips the reference's hard-coded [0, 0, 1] check
    with pytest.raises(ValueError):
        _plan([np.zeros((4, 5, 6))], np.zeros((3, 2, 2, 2)), affine=np.eye(4))


def test_public_api_fails_loudly_without_gpu_or_library():
    import torch
    import elasticdeform_amd as ed
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ed.deform_grid(np.zeros((8, 8)), np.zeros((2, 3, 3)))
    # argument errors still surface first, exactly like the reference
    with pytest.raises(AssertionError):
        ed.deform_grid(np.zeros((8, 8)), np.zeros((2, 3, 3)), order=7)
    with pytest.raises(ValueError, match="X_shape is required"):
        ed.deform_grid_gradient(np.zeros((4, 4)), np.zeros((2, 3, 3)), crop=(slice(0, 4),) * 2)


@pytest.mark.skipif(ref_loader.load_reference() is None,
                    reason="real reference only exists in the build container")
def test_plan_matches_live_reference_helpers():
    import importlib
    ref = ref_loader.load_reference()
    dg = importlib.import_module(ref.__name__ + ".deform_grid")
    rng = np.random.default_rng(1)
    X = rng.random((20, 30))
    for rotate, zoom in ((None, None), (17.0, None), (None, 0.7), (-33, 1.3)):
        for affine in (None, np.eye(2, 3) + rng.standard_normal((2, 3)) * 0.1):
            crop = (slice(2, 18), slice(5, 25))
            p = _plan([X], np.zeros((2, 3, 3)), crop=crop, affine=affine, rotate=rotate, zoom=zoom)
            Xs = dg._normalize_inputs(X)
            axis, ds = dg._normalize_axis_list(None, Xs)
            shapes, off = dg._compute_output_shapes(Xs, axis, ds, crop)
            inv = dg._compute_inverse_affine(dg._normalize_affine(affine, axis))
            inv = dg._apply_rotation_and_zoom(rotate, zoom, inv, [shapes[0][d] for d in axis[0]])
            np.testing.assert_array_equal(p.output_offset, off)
            if inv is None:
                assert p.inverse_affine is None
            else:
                np.testing.assert_array_equal(p.inverse_affine, inv)


def test_crop_window_logic():
    """Crop-aware prefilter: the host half -- tap window around the
    coordinate range, boundary handling, decay margin, 'is it worth it'."""
    X = np.zeros((3, 200, 210, 220), dtype=np.float32)
    disp = np.zeros((3, 3, 3, 3))
    plan = _host.Plan([X], disp, 3, 'constant', 0.0, (slice(80, 120),) * 3, [(1, 2, 3)], None, None, None)
    cbox = [(70, 130), (75, 125), (60, 140)]
    box = _host.source_box(plan, 0, X.shape, cbox)
    assert box == [(70 - 2, 130 + 3), (75 - 2, 125 + 3), (60 - 2, 140 + 3)]
    win = _host.prefilter_window(box, X.shape, (1, 2, 3), 3, slack_last=52)
    m = _host.PREFILTER_MARGIN[3]
    assert win == [(68 - m, 134 + m), (73 - m, 129 + m), ((58 - 52) & ~3, 144 + 52)]      # (rows start / end on multiples of 4)
    # a range that leaves the array: clipped for 'constant' / 'nearest' (keeping the mirror taps
    # of windows that stick out), whole axis for the folding modes
    cbox = [(-30, 20), (75, 125), (150, 260)]
    assert _host.source_box(plan, 0, X.shape, cbox) == [(0, 23), (73, 128), (148, 219)]
    cbox = [(-30, -10), (75, 125), (230, 260)]
    assert _host.source_box(plan, 0, X.shape, cbox) == [(0, 3), (73, 128), (216, 219)]
    plan_m = _host.Plan([X], disp, 3, 'mirror', 0.0, (slice(80, 120),) * 3, [(1, 2, 3)], None, None, None)
    assert _host.source_box(plan_m, 0, X.shape, cbox) == [(0, 199), (73, 128), (0, 219)]
    # a window that covers most of the volume is not worth a separate pass
    assert _host.prefilter_window([(0, 199), (0, 209), (10, 200)], X.shape, (1, 2, 3), 3) is None
    # even orders: floor(c + 0.5) - order // 2 .. + order
    plan2 = _host.Plan([X], disp, 2, 'constant', 0.0, (slice(80, 120),) * 3, [(1, 2, 3)], None, None, None)
    assert _host.source_box(plan2, 0, X.shape, [(70, 130)] * 3)[0] == (70 - 2, 130 + 2)


def test_gradient_entry_points_validate_before_touching_the_device():
    """deform_grid_gradient checks its arguments in the reference's order (deform_grid.py:246-266)
    BEFORE any allocation: the reference's exception classes come out even where no GPU exists."""
    import elasticdeform_amd as ed
    dY = np.zeros((9, 11), dtype=np.float32)
    with pytest.raises(AssertionError, match="Displacement matrix should be a numpy.ndarray"):
        ed.deform_grid_gradient(dY, [[0.0]])
    with pytest.raises(AssertionError, match="Displacement matrix should be a numpy.ndarray"):
        ed.deform_grid_gradient(dY, None)
    with pytest.raises(AssertionError, match="Number of dimensions of displacement"):
        ed.deform_grid_gradient(dY, np.zeros((2, 3)))
    with pytest.raises(AssertionError, match="order should be"):
        ed.deform_grid_gradient(dY, np.zeros((2, 3, 3)), order=7)
    with pytest.raises(ValueError, match="X_shape is required"):
        ed.deform_grid_gradient(dY, np.zeros((2, 3, 3)), crop=(slice(0, 4), slice(0, 4)))
    with pytest.raises(ValueError, match="X_shape does not match"):
        ed.deform_grid_gradient(dY, np.zeros((2, 3, 3)), crop=(slice(0, 4), slice(0, 4)),
                                X_shape=(9, 11))
    with pytest.raises(RuntimeError, match="boundary mode not supported"):
        ed.deform_grid_gradient(dY, np.zeros((2, 3, 3)), mode="bogus")
    with pytest.raises(AssertionError):
        ed.deform_grid_gradient_batch(np.zeros((2, 9, 11), np.float32), np.zeros((3, 2, 3, 3)))


def test_reference_import_name_is_served_by_the_alias_package():
    """`import elasticdeform` / `import elasticdeform.torch` (the reference's names,
    /root/reference/elasticdeform/__init__.py:1, torch.py:33) resolve to this build."""
    import elasticdeform
    import elasticdeform_amd
    assert elasticdeform.deform_grid is elasticdeform_amd.deform_grid
    assert elasticdeform.deform_random_grid is elasticdeform_amd.deform_random_grid
    assert elasticdeform.deform_grid_gradient is elasticdeform_amd.deform_grid_gradient
    torch = pytest.importorskip("torch")  # noqa: F841
    import elasticdeform.torch as etorch
    import elasticdeform_amd.torch as etorch_amd
    assert etorch.deform_grid is etorch_amd.deform_grid


def test_repeat_call_lane_only_recognises_device_tensors():
    """_fastlane.signature: anything that is not a CUDA tensor (list) on the current device -> None,
    i.e. the general path (no GPU needed to check that)."""
    torch = pytest.importorskip("torch")
    from elasticdeform_amd import _fastlane
    x = torch.zeros((4, 4))
    d = torch.zeros((2, 3, 3))
    args = (3, 'constant', 0.0, None, True, None, None, 0)
    assert _fastlane.signature(False, x, d, *args) is None
    assert _fastlane.signature(False, np.zeros((4, 4)), np.zeros((2, 3, 3)), *args) is None
    assert _fastlane.signature(False, [x], d, *args) is None
    assert _fastlane.signature(False, (x,), d, *args) is None
    assert _fastlane._hashable([3, 1]) == _fastlane._hashable([3, 1]) != _fastlane._hashable((3, 1))
    assert _fastlane._crop_key((slice(1, 5), slice(None))) == ((1, 5, None), (None, None, None))


def test_relayout_permutations_for_non_innermost_deformed_axes():
    """deform_grid._relayout_perms: which inputs are transposed to 'step axes first' (pure host logic)."""
    import importlib
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    S = _host.ShapeOnly
    disp = np.zeros((3, 3, 3, 3))

    def perms(shapes, axis, order=3):
        xs = [S(s) for s in shapes]
        return dgm._relayout_perms(_host.Plan(xs, disp, order, 'constant', 0.0, None, axis, None, None, None), xs)
    big = (64, 64, 64)
    assert perms([big], None) is None                                         # all axes deformed
    assert perms([(4,) + big], (1, 2, 3)) is None                             # channel-first: already trailing
    assert perms([big + (4,)], (0, 1, 2)) == [[3, 0, 1, 2]]                   # channel-last
    assert perms([(2,) + big + (3,)], (1, 2, 3)) == [[0, 4, 1, 2, 3]]         # step axes on both sides
    assert perms([big + (4,), (4,) + big], [(0, 1, 2), (1, 2, 3)]) == [[3, 0, 1, 2], None]
    assert perms([(8, 8, 8, 4)], (0, 1, 2)) is None                           # small: not worth two transposes
    assert perms([big + (4,)], (0, 1, 2), order=0) is None                    # order 0: one strided load either way
    assert perms([big + (4,), big + (4,)], (0, 1, 2), order=[0, 3]) == [None, [3, 0, 1, 2]]
    for p in ([3, 0, 1, 2], [0, 4, 1, 2, 3]):
        inv = dgm._inverse_perm(p)
        assert [p[i] for i in inv] == list(range(len(p))) and [inv[a] for a in p] == list(range(len(p)))


def test_sixteen_bit_volumes_stay_in_16_bits_only_where_the_tile_kernels_can_take_them():
    """deform_grid._direct16: the host's guess whether a 16-bit float volume can go through the float32 kernels
    without a cast pass (the library has the last word: EDHIP_ERR_UNSUPPORTED, nothing launched).  Opt-in only,
    never with 'exact' arithmetic, a crop, orders other than 2 / 3, lines outside 64..256, a last axis that is not a
    multiple of four samples, deformed axes that are not the three innermost, or float32 data."""
    torch = pytest.importorskip("torch")
    import importlib
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    x = torch.zeros((72, 96, 128), dtype=torch.bfloat16)
    ok = lambda t=x, axes=(0, 1, 2), order=3, prefilter=True, crop=None: dgm._direct16(t, axes, order, prefilter, crop)
    assert not ok()                                   # reduced precision is an opt-in
    prev = dgm.set_reduced_precision(True)
    try:
        assert ok() and ok(order=2) and ok(torch.zeros((2, 64, 64, 64), dtype=torch.float16), axes=(1, 2, 3))
        assert not ok(order=1) and not ok(order=4) and not ok(prefilter=False)
        assert not ok(crop=(slice(0, 8),) * 3)
        assert not ok(x.float())
        assert not ok(torch.zeros((72, 96, 130), dtype=torch.bfloat16))          # last axis not a multiple of 4
        assert not ok(torch.zeros((40, 96, 128), dtype=torch.bfloat16))          # a line below the tile kernels
        assert not ok(torch.zeros((72, 96, 260), dtype=torch.bfloat16))          # ... and above them
        assert not ok(torch.zeros((72, 96, 128, 4), dtype=torch.bfloat16), axes=(0, 1, 2))   # channel-last
        assert not ok(x.transpose(0, 1))                                        # not contiguous
        assert not ok(torch.zeros((96, 128), dtype=torch.bfloat16), axes=(0, 1))
        dgm.set_arithmetic("exact")
        assert not ok()
    finally:
        dgm.set_arithmetic("auto")
        dgm.set_reduced_precision(prev)


def test_a_deformed_axis_of_length_one_gives_the_reference_result():
    """The reference divides by (I - 1) = 0 for a deformed axis of length 1 (deform.c:643): every control coordinate is
    inf / NaN, every voxel maps to the constant in EVERY mode, the output is cval stored with the dtype's own rule
    (deform.c:287-306) and the gradient is zero.  Decided on the host (no GPU needed); against the oracle and, in the
    build container, the real reference."""
    import warnings
    import elasticdeform_amd as ed
    ref = ref_loader.load_reference()
    rng = np.random.default_rng(4)
    cases = [((1, 20), (3, 3), {}), ((12, 1, 9), (2, 3, 3), {}), ((1,), (3,), {}),
             ((3, 1, 7, 5), (2, 2, 2), dict(axis=(1, 2, 3))), ((9, 1), (3, 2), dict(crop=(slice(2, 7), slice(0, 1))))]
    for shape, pts, extra in cases:
        for dt in (np.float64, np.float32, np.uint8, np.int16, np.int32, np.uint16, bool):
            X = (rng.random(shape) * 50).astype(dt)
            disp = rng.standard_normal((len(pts),) + pts)
            for mode in ("constant", "nearest", "mirror", "reflect", "wrap"):
                for order, cval in ((0, 0.25), (1, -3.6), (3, 300.5)):
                    kw = dict(order=order, mode=mode, cval=cval, **extra)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        want = orc.deform_grid(X, disp, **kw)
                        if ref is not None:
                            np.testing.assert_array_equal(ref.deform_grid(X, disp, **kw), want)
                    got = ed.deform_grid(X, disp, **kw)
                    assert got.dtype == want.dtype and got.shape == want.shape
                    np.testing.assert_array_equal(got, want)
        # gradient: zero, in dY's dtype and X's shape; lists in, lists out
        kw = dict(order=3, mode="mirror", **extra)
        Xf = rng.random(shape)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = orc.deform_grid(Xf, disp, **kw)
            gw = orc.deform_grid_gradient(np.ones_like(out), disp, X_shape=shape, **kw)
        g = ed.deform_grid_gradient(np.ones_like(out), disp, X_shape=shape, **kw)
        assert g.shape == gw.shape == tuple(shape) and g.dtype == gw.dtype
        np.testing.assert_array_equal(g, gw)
        assert not g.any()
        both = ed.deform_grid([Xf, Xf.astype(np.float32)], disp, **kw)
        assert isinstance(both, list) and both[1].dtype == np.float32 and not both[0].any()
    with pytest.raises(RuntimeError):
        ed.deform_grid(np.zeros((1, 5), dtype=np.complex64), np.zeros((2, 2, 2)))
