"""
Argument normalisation of the public API: the GPU-independent half of one call.  Pure Python +
NumPy -- no torch, no HIP -- so it is testable on a machine without a GPU.

It reproduces what the reference's helpers (/root/reference/elasticdeform/deform_grid.py:295-454)
hand to ``_deform_grid.deform_grid(...)`` (deform_grid.py:174,274): per-input axis tuples, output
shapes and the int64 crop offset, int64 order / mode-code arrays, float64 cvals and the 2-D
inverse affine (user affine inverted, then composed with rotate / zoom about the centre of the
cropped output).  Failure behaviour follows the reference because its callers and tests rely on
it: AssertionError for inconsistent order / axis / displacement / crop, a plain Exception for an
X that is not an array (or list of arrays) and for a crop entry that is not a slice,
RuntimeError('boundary mode not supported') for an unknown mode name.

Arrays may be numpy.ndarray or torch.Tensor; only ``.shape`` / ``.ndim`` are inspected here.
"""
import functools

import numpy

# integer codes of the boundary modes: from_scipy.h:38-47 as used by deform_grid.py:440-454
MODE_CODES = {'nearest': 0, 'wrap': 1, 'reflect': 2, 'mirror': 3, 'constant': 4}


def is_array(x):
    """numpy.ndarray or torch.Tensor (duck-typed so that torch stays optional here)."""
    if isinstance(x, numpy.ndarray):
        return True
    return type(x).__module__.partition('.')[0] == 'torch' and hasattr(x, 'data_ptr')


def normalize_inputs(X):
    """array | list of arrays -> list of arrays (deform_grid.py:295-306)."""
    if is_array(X):
        return [X]
    if not isinstance(X, list):
        raise Exception('X should be a numpy.ndarray or a list of numpy.ndarrays.')
    assert len(X) > 0, 'You must provide at least one image.'
    assert all(is_array(x) for x in X), 'All elements of X should be numpy.ndarrays.'
    return X


def normalize_axis_list(axis, Xs):
    """None | int | tuple | list of tuples -> (list of per-input tuples, deformed shape)
    (deform_grid.py:308-326)."""
    if axis is None:
        per_input = [tuple(range(x.ndim)) for x in Xs]
    else:
        if isinstance(axis, int):
            axis = (axis,)
        per_input = [axis] * len(Xs) if isinstance(axis, tuple) else axis
    assert len(per_input) == len(Xs), 'Number of axis tuples should match number of inputs.'
    naxis = None
    shapes = set()
    for x, ax in zip(Xs, per_input):
        assert isinstance(ax, tuple), 'axis should be given as a tuple'
        assert all(isinstance(a, int) for a in ax), 'axis must contain ints'
        naxis = len(ax) if naxis is None else naxis
        assert len(ax) == naxis, 'All axis tuples should have the same length.'
        # NB the reference compares with tuple(set(ax)); for small non-negative ints that is
        # "strictly ascending", and negative axes are rejected by the range check below
        assert ax == tuple(set(ax)), 'axis must be sorted and unique'
        assert all(0 <= a < x.ndim for a in ax), 'invalid axis for input'
        shapes.add(tuple(int(x.shape[a]) for a in ax))
    assert len(shapes) == 1, 'All inputs should have the same shape.'
    return per_input, shapes.pop()


def compute_output_shapes(Xs, axis, deform_shape, crop):
    """crop (sequence of plain slices over the deformed axes) -> per-input output shapes and the
    int64 offset vector, which is None unless some start > 0 (deform_grid.py:328-354)."""
    if crop is None:
        return [tuple(x.shape) for x in Xs], None
    assert isinstance(crop, (tuple, list)), "crop must be a tuple or a list."
    assert len(crop) == len(deform_shape)
    shapes = [list(x.shape) for x in Xs]
    starts = []
    for d, (sl, full) in enumerate(zip(crop, deform_shape)):
        if not isinstance(sl, slice):
            raise Exception('Crop must be a slice.')
        assert sl.step is None
        start, stop = sl.start or 0, sl.stop or full
        assert start >= 0
        assert start < stop and stop <= full
        for shape, ax in zip(shapes, axis):
            shape[ax[d]] = stop - start
        starts.append(start)
    offset = numpy.array(starts).astype('int64') if any(s > 0 for s in starts) else None
    return shapes, offset


def check_displacement(displacement, naxis):
    """deform_grid.py:356-360"""
    assert is_array(displacement), 'Displacement matrix should be a numpy.ndarray.'
    assert displacement.ndim == naxis + 1, \
        'Number of dimensions of displacement does not match input.'
    assert displacement.shape[0] == naxis, \
        'First dimension of displacement should match number of input dimensions.'


def _per_input(value, n, what):
    values = list(value) if isinstance(value, (tuple, list)) else [value] * n
    assert len(values) == n, \
        'Number of %s parameters should be equal to number of inputs.' % what
    return values


def normalize_order(order, n):
    """-> int64[n], each 0..5 (deform_grid.py:362-367; the docstring there says 0-4, the code
    accepts 5)."""
    orders = _per_input(order, n, 'order')
    assert all(0 <= o and o <= 5 for o in orders), 'order should be 0, 1, 2, 3, 4 or 5.'
    return numpy.array(orders).astype('int64')


def mode_code(mode):
    try:
        return MODE_CODES[mode]
    except (KeyError, TypeError):
        raise RuntimeError('boundary mode not supported')


def normalize_mode(mode, n):
    """-> int64[n] of mode codes (deform_grid.py:369-374)."""
    if isinstance(mode, (tuple, list)):
        codes = [mode_code(m) for m in mode]       # unknown names fail before the length check,
    else:                                          # as in the reference
        codes = [mode_code(mode)] * n
    assert len(codes) == n, 'Number of mode parameters should be equal to number of inputs.'
    return numpy.array(codes).astype('int64')


def normalize_cval(cval, n):
    """-> float64[n] (deform_grid.py:376-380)"""
    return numpy.array(_per_input(cval, n, 'cval')).astype('float64')


def inverse_of_affine(affine, naxis):
    """User affine (naxis x naxis+1, or a homogeneous 3x3 in 2-D) -> float64 inverse map
    [M^-1 | -M^-1 b], or None (deform_grid.py:382-399)."""
    if affine is None:
        return None
    if hasattr(affine, 'detach'):                    # torch.Tensor convenience
        affine = affine.detach().cpu().numpy()
    affine = numpy.asarray(affine)
    if affine.shape == (naxis + 1, naxis + 1):
        # the reference hard-codes the 2-D bottom row, so a homogeneous matrix only passes in 2-D
        assert numpy.allclose(affine[naxis, :], [0, 0, 1]), 'Invalid affine matrix.'
        affine = affine[:naxis, :]
    assert affine.shape == (naxis, naxis + 1), 'Affine matrix should have shape (ndim, ndim+1).'
    affine = numpy.array(affine).astype('float64')
    inv = numpy.zeros(affine.shape, dtype='float64')
    inv[:, :-1] = numpy.linalg.inv(affine[:, :-1])
    inv[:, -1] = -numpy.dot(inv[:, :-1], affine[:, -1])
    return inv


def _shift(c, sign):
    return numpy.array([[1, 0, sign * c[0]], [0, 1, sign * c[1]], [0, 0, 1]])


def compose_rotation_zoom(rotate, zoom, inverse_affine, out_deform_shape):
    """rotate (degrees) / zoom about the centre of the cropped output, left-multiplied onto the
    inverse affine.  2-D only (deform_grid.py:401-438).  The factors are multiplied in the
    reference's order -- T(-c), then R, then Z, then T(c), each from the left -- so the float64
    matrix is the same to the last bit."""
    if rotate is None and zoom is None:
        return inverse_affine
    assert len(out_deform_shape) == 2, 'Zoom and rotate is only implemented for 2D images.'
    angle = -float(rotate or 0)
    scale = 1 / float(zoom or 1)
    centre = numpy.array(out_deform_shape) / 2 - 0.5
    factors = [_shift(centre, -1)]
    if angle:
        th = numpy.radians(angle)
        factors.append(numpy.array([[numpy.cos(th), -numpy.sin(th), 0],
                                    [numpy.sin(th), numpy.cos(th), 0],
                                    [0, 0, 1]]))
    if scale:
        factors.append(numpy.array([[scale, 0, 0], [0, scale, 0], [0, 0, 1]]))
    factors.append(_shift(centre, +1))
    m = functools.reduce(lambda acc, f: numpy.dot(f, acc), factors)
    if inverse_affine is None:
        return m[:2, :]
    base = numpy.eye(3, dtype='float64')
    base[:-1, :] = inverse_affine
    return numpy.dot(m, base)[:2, :]


_INT_LIMITS = {'uint8': (0.0, 255.0), 'uint16': (0.0, 65535.0), 'uint32': (0.0, 4294967295.0),
               'uint64': (0.0, 18446744073709551615.0), 'int8': (-128.0, 127.0), 'int16': (-32768.0, 32767.0),
               'int32': (-2147483648.0, 2147483647.0), 'int64': (-9223372036854775808.0, 9223372036854775807.0)}


def stored_constant(cval, dtype_name):
    """The value the reference stores for a voxel that maps to the constant (deform.c:287-306,906-919): float types take
    the C cast of ``cval``; signed integers round half away from zero and clamp; unsigned ones add 0.5 to positives,
    send everything else to 0, clamp; bool takes the C cast to unsigned char.  Returns a Python scalar for numpy.full /
    torch.full of that dtype."""
    t = float(cval)
    if dtype_name == 'bool':
        return bool(int(t) & 0xFF) if t == t and abs(t) < 2.0 ** 31 else False
    if dtype_name in _INT_LIMITS:
        lo, hi = _INT_LIMITS[dtype_name]
        if not t == t:
            return 0
        if lo < 0:
            t = t + 0.5 if t > 0 else t - 0.5
        else:
            t = t + 0.5 if t > 0 else 0.0
        t = min(max(t, lo), hi)
        return max(int(lo), min(int(t), int(hi) if hi < 9e18 else (2 ** 64 - 1 if lo == 0 else 2 ** 63 - 1)))
    return t


def degenerate_axis(shapes, axes):
    """True when a deformed axis of an input has length 1: the reference divides by (I - 1) = 0 there (deform.c:643),
    every control coordinate is inf or NaN, every voxel maps to the constant in every mode -- the result is ``cval``
    everywhere and the gradient is zero (checked against the real reference: tests/test_oracle.py)."""
    return any(int(sh[a]) == 1 for sh, ax in zip(shapes, axes) for a in ax)


class ShapeOnly(object):
    """Stand-in for an array of which only the shape is known (the dX of a gradient call before it
    is allocated): Plan inspects ``.shape`` / ``.ndim`` only."""
    __slots__ = ("shape", "ndim")

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in shape)
        self.ndim = len(self.shape)


class Plan(object):
    """The normalised, array-free part of one deform_grid / deform_grid_gradient call -- i.e. the
    non-array arguments of _deform_grid.deform_grid (_deform_grid.c:108-118)."""

    __slots__ = ("axis", "naxis", "deform_shape", "output_shapes", "output_offset", "order",
                 "mode", "cval", "inverse_affine", "prepared")

    def __init__(self, Xs, displacement, order, mode, cval, crop, axis, affine, rotate, zoom):
        # same order of checks as deform_grid.py:135-152 / :246-266
        n = len(Xs)
        self.axis, self.deform_shape = normalize_axis_list(axis, Xs)
        self.naxis = len(self.axis[0])
        self.output_shapes, self.output_offset = compute_output_shapes(
            Xs, self.axis, self.deform_shape, crop)
        check_displacement(displacement, self.naxis)
        self.order = normalize_order(order, n)
        self.mode = normalize_mode(mode, n)
        self.cval = normalize_cval(cval, n)
        inv = inverse_of_affine(affine, self.naxis)
        self.inverse_affine = compose_rotation_zoom(
            rotate, zoom, inv, [self.output_shapes[0][d] for d in self.axis[0]])
        self.prepared = None         # ctypes form of the arrays above, attached by the caller


# ---- plan cache ------------------------------------------------------------------------------------
# Normalising the arguments of one call costs ~15 us of Python (more than a small volume's kernels);
# training loops repeat the same call with new data, so plans are memoised on everything they depend
# on: shapes and the non-array arguments.  Only successfully validated plans are stored.
_PLAN_CACHE = {}
_PLAN_CACHE_MAX = 128


def _freeze(v):
    if isinstance(v, (list, tuple)):
        return (type(v).__name__,) + tuple(_freeze(x) for x in v)
    if isinstance(v, slice):
        return ('slice', v.start, v.stop, v.step)
    if isinstance(v, numpy.ndarray):
        return ('nd', v.shape, str(v.dtype), v.tobytes())
    if hasattr(v, 'detach') and hasattr(v, 'cpu'):        # torch.Tensor (affine convenience)
        a = v.detach().cpu().numpy()
        return ('nd', a.shape, str(a.dtype), a.tobytes())
    if isinstance(v, (int, float, str, bool, type(None))):
        return (type(v).__name__, v)
    raise TypeError('unhashable plan argument')


def cached_plan(Xs, displacement, order, mode, cval, crop, axis, affine, rotate, zoom):
    """Plan(...) memoised on (input shapes, displacement shape, every non-array argument)."""
    try:
        key = (tuple(tuple(int(d) for d in x.shape) for x in Xs),
               tuple(int(d) for d in displacement.shape) if is_array(displacement) else None,
               _freeze(order), _freeze(mode), _freeze(cval), _freeze(crop), _freeze(axis),
               _freeze(affine), _freeze(rotate), _freeze(zoom))
        hit = _PLAN_CACHE.get(key)
    except (TypeError, AttributeError):
        return Plan(Xs, displacement, order, mode, cval, crop, axis, affine, rotate, zoom)
    if hit is not None:
        return hit
    plan = Plan(Xs, displacement, order, mode, cval, crop, axis, affine, rotate, zoom)
    if len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
        _PLAN_CACHE.clear()
    _PLAN_CACHE[key] = plan
    return plan


# ---- crop-aware prefilter --------------------------------------
# With a crop only a box of the source volume is ever read.  The B-spline prefilter is an
# exponentially decaying filter (|pole|^k), so filtering that box plus a margin gives the same
# coefficients inside the box as filtering the whole volume, to below fp64 rounding.

# margin (samples) after which a cut in the line is invisible: |z|^m < 1e-18 for the largest pole
PREFILTER_MARGIN = {2: 24, 3: 32, 4: 56, 5: 64}


def source_box(plan, i, in_shape, cbox):
    """Inclusive index range [lo, hi] along every deformed axis of input `i` that contains every
    source sample a tap of the output can touch.  `cbox[h] = (floor(min c_h), ceil(max c_h))` is
    the range of the source coordinate before the boundary map (edhip_source_box); the tap window
    of deform.c:783-813 is added here.  Where the range leaves the array the boundary map
    decides: 'nearest' / 'constant' clip it, the folding modes fall back to the whole axis."""
    ax = plan.axis[i]
    order = int(plan.order[i])
    mode = int(plan.mode[i])
    box = []
    for h in range(plan.naxis):
        n = int(in_shape[ax[h]])
        # one sample of slack: the kernels' own coordinate arithmetic may round differently
        lo = int(cbox[h][0]) - order // 2 - 1
        hi = int(cbox[h][1]) + order - order // 2 + 1
        if lo < 0 or hi > n - 1:
            if mode in (MODE_CODES['nearest'], MODE_CODES['constant']):
                clipped_lo, clipped_hi = lo < 0, hi > n - 1
                lo, hi = min(max(lo, 0), n - 1), max(min(hi, n - 1), 0)
                # windows that stick out are mirror-indexed (deform.c:795-813): up to `order`
                # samples inward of the edge
                if clipped_lo:
                    hi = max(hi, min(order, n - 1))
                if clipped_hi:
                    lo = min(lo, max(n - 1 - order, 0))
            else:
                lo, hi = 0, n - 1
        box.append((lo, hi))
    return box


def prefilter_window(box, in_shape, axes, order, slack_last=0):
    """The box grown by the filter's decay margin, clipped to the array: slices over the deformed
    axes, or None when that would not save at least 40 % of the volume.  `slack_last` extra
    samples are kept after the box on the last axis (the staging loads of the tile kernels read
    whole padded rows)."""
    m = PREFILTER_MARGIN.get(int(order), 64)
    win = []
    sub = full = 1
    for k, ((lo, hi), a) in enumerate(zip(box, axes)):
        n = int(in_shape[a])
        extra = slack_last if k == len(box) - 1 else 0
        w0, w1 = max(0, lo - max(m, extra)), min(n, hi + 1 + max(m, extra))
        if k == len(box) - 1:
            # rows of the window start and end on 16-byte boundaries of the array's rows, so that the
            # whole-line tile kernels move them as vectors (an odd start sent every pass to the scalar variants)
            w0, w1 = w0 & ~3, min(n, (w1 + 3) & ~3)
        win.append((w0, w1))
        sub *= w1 - w0
        full *= n
    if sub > 0.6 * full:
        return None
    return win
