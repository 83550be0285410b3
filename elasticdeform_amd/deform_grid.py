"""
Public API: ``deform_random_grid``, ``deform_grid``, ``deform_grid_gradient`` -- drop-in for
``elasticdeform.deform_random_grid / deform_grid / deform_grid_gradient``
(/root/reference/elasticdeform/deform_grid.py:6-8, :52-53, :182-184): same names, positional
order, defaults, list-in => list-out, exception classes.

What differs is where the work happens: every array is (moved to) MI355X HBM and the whole
pipeline -- B-spline prefilter of the inputs and of the displacement grid, the per-voxel
deformation, and for the gradient the scatter-add plus the transposed prefilter -- runs as
hand-written HIP kernels behind the C ABI of include/edhip.h, enqueued on the current HIP stream
with no host synchronisation (one exception, documented in include/edhip.h: the first call on a
stream that needs more scratch than that stream's cached workspace holds waits for the stream once).

Limits that differ from the reference (status EDHIP_ERR_UNSUPPORTED / EDHIP_ERR_INVALID ->
RuntimeError): at most 8 array dimensions, hence at most 7 deformed axes (the control grid has one
dimension more; the reference takes whatever NumPy does, _deform_grid.c:158-175).  A deformed axis of length 1 gives what
the reference gives -- it divides by I - 1 = 0 there (deform.c:643), every coordinate is inf / NaN and maps to cval: the
output is cval everywhere, the gradient zero -- decided on the host (the C ABI itself refuses such an axis).

* numpy.ndarray in  -> numpy.ndarray out (one H2D and one D2H copy; the drop-in path)
* torch.Tensor in   -> torch.Tensor out on the same device (CUDA tensors never touch the host)

There is no CPU fallback: without a GPU or without the built library the call raises.
"""
import os
import sys
import threading

import warnings

import numpy

from . import _fastlane
from . import _host
from . import _lib

_this = sys.modules[__name__]

_ARITHMETIC = {'auto': _lib.FLAG_AUTO, 'exact': _lib.FLAG_EXACT, 'fast': _lib.FLAG_FAST}
_flags = _ARITHMETIC.get(os.environ.get('EDHIP_ARITHMETIC', 'auto').lower(), _lib.FLAG_AUTO)


def set_arithmetic(kind):
    """Select the kernels' arithmetic: 'auto' (float32 / float64 volumes -> fast path, integer and
    bool volumes -> exact path), 'exact' (fp64 arithmetic in the reference's evaluation order for
    every dtype: float64 / float32 / integer outputs bit-comparable with the reference) or 'fast'
    (same as 'auto' today).  Returns the previous value."""
    global _flags
    if kind not in _ARITHMETIC:
        raise ValueError("arithmetic must be one of %s" % sorted(_ARITHMETIC))
    prev = [k for k, v in _ARITHMETIC.items() if v == _flags][0]
    _flags = _ARITHMETIC[kind]
    return prev


_reduced = os.environ.get('EDHIP_REDUCED_PRECISION', '') not in ('', '0')


def set_reduced_precision(enabled):
    """Opt in to reduced-precision I/O (off by default, where
    float16 input raises 'data type not supported' exactly like the reference, deform.c:742-747):

    * float16 / bfloat16 volumes (torch tensors; numpy float16): stored in 16 bits, computed in
      float32 -- prefilter, interpolation and the gradient's scatter all run on the widened values
      and only the final result is rounded (to nearest even) to the storage type.  Result =
      round_storage(float32 pipeline(widen(X))).
    * integer images with order > 1: the reference prefilters them INTO their integer dtype
      (deform_grid.py:158), which destroys the interpolation (off by up to 225 grey
      levels on uint8).  With the opt-in the prefilter and the interpolation run in float32 and the
      result is stored with the reference's own rounding / clamping rule (deform.c:292-306).  This
      deliberately DIFFERS from the reference for such images; order 0 / 1 are unaffected.

    With arithmetic 'exact' 16-bit volumes go through the fp64 kernels in their storage type
    (rounded after every filter axis, like every dtype there).  Returns the previous setting."""
    global _reduced
    prev = _reduced
    _reduced = bool(enabled)
    return prev


_strict_crop = os.environ.get('EDHIP_STRICT_CROP', '') not in ('', '0')


def set_crop_identity(strict):
    """The reference's crop guarantee, bit for bit (README.md:113 ``full[crop] == cropped``): with ``strict=True`` the
    spline prefilter of a cropped call always runs over the WHOLE input, exactly as in the uncropped call, so the
    cropped result equals the same region of the full result in every bit.  Off by default: large inputs with a small
    crop then prefilter only the part of the volume the crop can reach (plus a decay margin), which changes float32 /
    float64 results by ~1e-9 of the data's scale -- far inside every tolerance, but not identical bits -- and saves
    most of the prefilter (BASELINE cfg4: forward 263 -> 140 us).  Integer volumes and 'exact' arithmetic never use
    the window.  Returns the previous setting."""
    global _strict_crop
    prev = _strict_crop
    _strict_crop = bool(strict)
    return prev


_FIELD_STRENGTH = ('auto', 'strong')
_field_strength = os.environ.get('EDHIP_FIELD_STRENGTH', 'auto').lower()
if _field_strength not in _FIELD_STRENGTH:
    _field_strength = 'auto'


def set_field_strength(kind):
    """A performance hint for the forward kernels of float32 volumes with three deformed axes (results agree to float32
    rounding either way).  'strong' says the displacement fields are strong -- a displacement gradient of ~0.15 per voxel
    and more, e.g. sigma >= 10 on a 5^3 grid over 256^3, or the reference README's ``sigma=25, points=3`` on a 200 x 300
    image: the z-walk kernels (tiles split in halves, wide row pitches) then serve every geometry they support and are
    5-35 % faster on such fields; on mild fields they are 5-50 % slower than the default choice for small volumes and
    batches (profiles/r06_k1_route_sweep.txt).  'auto' (default) picks by the volume's shape alone.  The library never
    picks by looking at earlier calls: a call's bits depend on its arguments and on this setting only.  Returns the
    previous value."""
    global _field_strength
    if kind not in _FIELD_STRENGTH:
        raise ValueError("field strength must be one of %s" % list(_FIELD_STRENGTH))
    prev = _field_strength
    _field_strength = kind
    return prev


_tls = threading.local()
# deform_random_grid knows what the library cannot see without a host round trip: sigma and the control-point spacing.
# sigma * (points - 1) / (extent - 1) is the displacement gradient's scale; the z-walk kernels win from ~0.15
# (5 points: sigma 10 at 256^3 = 0.157: 128^3 +23 %, 192^3 +5 %, 256^3 +19 %, batches even; sigma 15: +20 ... +27 %; sigma 5:
# -7 ... -23 % -- profiles/r06_k1_route_sweep.txt, last table)
RANDOM_GRID_STRONG_FROM = 0.15


class _random_grid_hint(object):
    """Context of one deform_random_grid call: the forward call inside it carries EDHIP_FLAG_STRONG_FIELD when the
    drawn field is strong by construction.  A function of the call's own arguments (sigma, points, shape)."""

    def __init__(self, sigma, points, deform_shape):
        try:
            self.strong = any(abs(float(sigma)) * (int(p) - 1) >= RANDOM_GRID_STRONG_FROM * (int(n) - 1)
                              for p, n in zip(points, deform_shape) if int(n) > 1)
        except (TypeError, ValueError):
            self.strong = False

    def __enter__(self):
        self.prev = getattr(_tls, 'strong', False)
        _tls.strong = self.strong
        return self

    def __exit__(self, *exc):
        _tls.strong = self.prev
        return False


def _route_flags():
    return _lib.FLAG_STRONG_FIELD if (_field_strength == 'strong' or getattr(_tls, 'strong', False)) else 0


_GRAD_ACCUMULATION = ('fixed', 'float')
_grad_accumulation = os.environ.get('EDHIP_GRAD_ACCUMULATION', 'fixed').lower()
if _grad_accumulation not in _GRAD_ACCUMULATION:
    _grad_accumulation = 'fixed'


def set_gradient_accumulation(kind):
    """How ``deform_grid_gradient`` sums the taps of float32 / float64 gradients (deform.c:926-997 adds every tap
    into the output array in its own precision):

    * 'fixed' (default): the tile kernels accumulate in fixed-point LDS cells whose scale comes from the sum of |dY|
      over the tile (include/edhip.h).  Every contribution is resolved to ~1.4e-10 of its TILE's sum of |dY|: exact
      to float32 rounding for gradients of ordinary dynamic range, but one 1e6 outlier costs the other voxels of its
      8 x 8 x 16 tile ~1e-4 absolute.
    * 'float': every tap is added with a floating-point atomic in the array's own type, as the reference does -- the
      precision of a contribution is relative to the contribution itself, whatever its neighbours hold.  The scatter
      runs on the one-thread-per-voxel kernel with the reference's fp64 coordinate arithmetic (several times slower
      than the tile kernels); prefilter transposes are unchanged.

    Returns the previous setting."""
    global _grad_accumulation
    if kind not in _GRAD_ACCUMULATION:
        raise ValueError("gradient accumulation must be one of %s" % list(_GRAD_ACCUMULATION))
    prev = _grad_accumulation
    _grad_accumulation = kind
    return prev


def _torch():
    import torch
    return torch


_TORCH_NAMES = None


def _dtype_name(t):
    """torch dtype -> the NumPy-style name used by the C ABI table."""
    global _TORCH_NAMES
    if _TORCH_NAMES is None:
        torch = _torch()
        _TORCH_NAMES = {getattr(torch, n): n for n in _lib.DTYPE_CODES if hasattr(torch, n)}
    name = _TORCH_NAMES.get(t.dtype)
    if name is None or (name in _lib.REDUCED_DTYPES and not _reduced):
        raise RuntimeError('data type not supported')     # deform.c:744,891 (float16, complex ...)
    return name


def _device_for(arrays):
    """The GPU every array of this call lives on / is moved to.  Fails loudly without one."""
    torch = _torch()
    for a in arrays:
        if not isinstance(a, numpy.ndarray) and a.is_cuda:
            return a.device
    if not torch.cuda.is_available():
        raise RuntimeError('elasticdeform_amd needs a ROCm GPU (MI355X / gfx950): none is visible '
                           'and there is no CPU fallback.')
    return torch.device('cuda', torch.cuda.current_device())


def _to_device(x, device):
    """numpy / torch array -> tensor in HBM on `device`, keeping the logical strides when the
    bridge allows it (the reference accepts arbitrary strides, _deform_grid.c:12-15)."""
    torch = _torch()
    if isinstance(x, numpy.ndarray):
        if x.dtype.name not in _lib.DTYPE_CODES or (x.dtype.name in _lib.REDUCED_DTYPES and not _reduced):
            raise RuntimeError('data type not supported')
        if not x.dtype.isnative or not x.flags.aligned or any(s < 0 for s in x.strides) \
                or not x.flags.writeable:
            x = numpy.ascontiguousarray(x).astype(x.dtype.newbyteorder('='), copy=True)
        return torch.from_numpy(x).to(device)
    x = x.detach()
    if x.device != device:
        x = x.to(device)
    return x


def _from_device(t, like):
    """Give the result back in the caller's array family."""
    if isinstance(like, numpy.ndarray):
        return t.cpu().numpy()
    if like.device != t.device:
        return t.to(like.device)
    return t


def _constant_result(like, shape, cval):
    """An array of `shape` in the family, dtype and device of `like`, filled with the value the reference stores for
    a constant voxel (_host.stored_constant)."""
    shape = tuple(int(v) for v in shape)
    if isinstance(like, numpy.ndarray):
        name = like.dtype.name
        if name not in _lib.DTYPE_CODES or (name in _lib.REDUCED_DTYPES and not _reduced):
            raise RuntimeError('data type not supported')     # deform.c:744,891
        return numpy.full(shape, _host.stored_constant(cval, name), dtype=like.dtype)
    torch = _torch()
    return torch.full(shape, _host.stored_constant(cval, _dtype_name(like)), dtype=like.dtype, device=like.device)


_INT_RANGE = {'uint8': (0.0, 255.0), 'uint16': (0.0, 65535.0), 'uint32': (0.0, 4294967295.0),
              'int8': (-128.0, 127.0), 'int16': (-32768.0, 32767.0), 'int32': (-2147483648.0, 2147483647.0)}


def _widen(t, order, prefilter):
    """Reduced-precision opt-in: the float32 stand-in of a 16-bit float volume, or of an integer
    image that would lose its prefilter to integer rounding; None when `t` runs as it is."""
    if not _reduced or (_flags & _lib.FLAG_EXACT):
        return None
    name = _TORCH_NAMES.get(t.dtype) if _TORCH_NAMES else None
    if name is None:
        name = _dtype_name(t)
    if name in _lib.REDUCED_DTYPES or (name in _INT_RANGE and order > 1 and prefilter):
        return t.to(_torch().float32)
    return None


def _direct16(x, axes, order, prefilter, crop):
    """Reduced-precision opt-in, 16-bit float volume: can it stay in 16 bits in HBM?  (The float32 kernels then
    widen it where they read it and narrow where they write: the first prefilter pass, K1's store, K2's load of
    dY, the last transposed prefilter pass -- no cast passes, 16 instead of 44 bytes of casts + I/O per voxel
    around the float32 intermediates.)  This is only the host's guess from shapes; the library has the last word
    (EDHIP_ERR_UNSUPPORTED, nothing launched) and the caller then falls back to widening with a cast."""
    if not _reduced or (_flags & _lib.FLAG_EXACT) or crop is not None or not prefilter or order not in (2, 3):
        return False
    torch = _torch()
    if x.dtype not in (torch.float16, torch.bfloat16) or not x.is_contiguous() or len(axes) != 3 or x.dim() < 3:
        return False
    if tuple(axes) != tuple(range(x.dim() - 3, x.dim())) or int(x.shape[-1]) % 4:
        return False
    return all(64 <= int(x.shape[a]) <= 256 for a in axes)


def _narrow(t32, like):
    """float32 result -> the storage dtype of `like`: round to nearest even for 16-bit floats; the
    reference's store rule for integers (deform.c:292-306: round half away from zero, clamp)."""
    torch = _torch()
    name = _dtype_name(like)
    if name in _lib.REDUCED_DTYPES:
        return t32.to(like.dtype)
    lo, hi = _INT_RANGE[name]
    # in float64 like the reference (`t` is a double there): 0.49999997f + 0.5f rounds up to 1.0f in
    # float32 and would store 1 where the double rule stores 0
    t = t32.double()
    nan = torch.isnan(t)
    r = torch.where(t > 0, t + 0.5, t - 0.5) if lo < 0 else torch.where(t > 0, t + 0.5, torch.zeros_like(t))
    r = r.clamp_(lo, hi).trunc_()
    # NaN passes every comparison of the store rule and reaches the cast: x86-64's cvttsd2si gives the
    # "integer indefinite" 0x80...0, of which the narrow and the unsigned types keep the low bits (0)
    # -- what the exact kernels' store_forward produces
    r = torch.where(nan, torch.full_like(r, float(lo) if name in ('int32', 'int64') else 0.0), r)
    return r.to(like.dtype)


def _desc(t):
    es = t.element_size()
    return _lib.describe(t.data_ptr(), _dtype_name(t), t.shape, [s * es for s in t.stride()])


def _prepared(plan, n):
    """ctypes form of the plan's parameter arrays, built once per (cached) plan"""
    if plan.prepared is None:
        plan.prepared = _lib.DeformArgs(n, plan.axis, plan.order, plan.mode, plan.cval, plan.output_offset,
                                        plan.inverse_affine)
    return plan.prepared


def _desc_sample0(t):
    """Descriptor of sample 0 of a stacked tensor (B, ...) and the byte distance between samples."""
    es = t.element_size()
    return (_lib.describe(t.data_ptr(), _dtype_name(t), t.shape[1:], [s * es for s in t.stride()[1:]]),
            t.stride(0) * es)


def _stream(device):
    return _torch().cuda.current_stream(device).cuda_stream


# Forward -> gradient hand-over of the tile bounding boxes (EDHIP_FLAG_KEEP_BOXES / USE_BOXES): per
# (device, stream), the storage address and version counter of the displacement tensor of the last
# forward call.  The gradient call sets USE_BOXES when it is handed the same storage, unmodified
# (autograd's backward, or deform_grid_gradient after deform_grid in a training step).  This is a
# heuristic about CONTENTS only: the library checks every other argument itself and treats the boxes
# as a hint, so a tensor changed behind PyTorch's back (`.data`), or a new tensor that landed on a
# freed one's address, costs time and never correctness.
_box_owner = {}
_batch_grids = {}      # (device, stream) -> (identity of the displacement tensor, its filtered grids) of the last batch forward


def _box_id(displacement, df):
    torch = _torch()
    if torch.is_tensor(displacement) and displacement.is_cuda and df.data_ptr() == displacement.data_ptr():
        return (displacement.data_ptr(), displacement._version)
    return None


def _box_flag_forward(displacement, df, device, stream):
    """KEEP_BOXES for a forward call whose control grid is the caller's own device tensor."""
    ident = _box_id(displacement, df)
    _box_owner[(device.index, stream)] = ident
    _batch_grids[(device.index, stream)] = None        # (the stream's box buffer changes hands)
    return _lib.FLAG_KEEP_BOXES if ident is not None else 0


def _box_flag_gradient(displacement, df, device, stream):
    ident = _box_id(displacement, df)
    if ident is not None and _box_owner.get((device.index, stream)) == ident:
        return _lib.FLAG_USE_BOXES
    return 0


def _filter_axes(x, axes, order, transpose, device, overwrite=False, stream=None):
    """Chain of 1-D spline filters over `axes` -- the reference's loop at deform_grid.py:157-162
    (forward) / :279-284 (transpose).  The reference filters x -> x_f and then x_f in place; so does
    this chain for lines of up to 256 samples, and it ping-pongs between two buffers for longer ones
    (the caller's x is never written); the values are the same."""
    torch = _torch()
    axes = list(axes)
    if not axes:
        return x
    if stream is None:
        stream = _stream(device)
    # Lines that fit the whole-line tile kernels are filtered in place from the second pass on
    # (like the reference; one temporary instead of two keeps the step's working set smaller);
    # longer lines ping-pong, because in place the block-recompute kernels cannot split a line.
    inplace = all(int(x.shape[d]) <= 256 for d in axes)
    if overwrite and inplace:
        bufs = [x, None]            # x is the caller's own temporary (dX): every pass in place
    else:
        bufs = [torch.empty_like(x), torch.empty_like(x) if (len(axes) > 1 and not inplace) else None]
    if inplace:
        # one library call for the whole chain: x -> bufs[0], then in place
        _lib.spline_filter_axes(_desc(x), _desc(bufs[0]), axes, order, transpose, _flags, stream)
        return bufs[0]
    src = x
    for i, d in enumerate(axes):
        dst = bufs[i & 1]
        _lib.spline_filter1d(_desc(src), _desc(dst), d, order, transpose, _flags, stream)
        src = dst
    return src


# Crop-aware prefilter: with a crop only a box of each input is ever read, so only that
# box plus the filter's decay margin is filtered.  The window never leaves the device: edhip_source_window
# computes it from the control grid (one small launch, no read-back), the filter passes and the transposed
# passes of the gradient restrict themselves to it (edhip_spline_filter_axes_window), K1 / K2 work on full-size
# buffers whose samples outside the window are never touched.  The host only decides WHETHER to try, from what
# it knows without the device: the output box plus the margins, as a fraction of the input.
# (Until round 4 the box was read back to the host: the synchronisation exposed the launch latency of every
# kernel behind it, and BASELINE cfg4 -- 3 x 256^3 cropped to 64^3 -- never got a window.)
CROP_WINDOW_MIN_SAVING = 24e6        # voxels per filter pass, summed over the inputs: below, the window's own launches and
                                     # host calls (~30 us) eat what it saves (a 256^3 float32 pass is 25 us; profiles/r04_time_crop_window.txt)
CROP_WINDOW_MAX_FRACTION = 0.50      # (output box + margins) / input, upper bound known to the host
# decay margin (samples) after which a cut in a line is invisible in the data's own precision: |pole|^m below
# 1e-9 for float32 volumes (their filter runs in float32), below 1e-18 for float64
_WINDOW_MARGIN = {'float32': {2: 12, 3: 16}, 'float64': {2: 24, 3: 32}}
_WINDOW_MIN_LINE = 64                # the whole-line tile kernels' shortest line


def _crop_window_pays(plan, shapes, names, todo, in_len, out_len, disp_shape=None, contiguous=None):
    """What the host knows without the device: does the smallest possible window (output box + margins + taps, at
    least a tile kernel's shortest line) save enough filter work, summed over the inputs `todo`?  One rule for the
    general path (_crop_windows) and for the repeat-call lanes (_fastlane.Lane.window_pays), so that repeated
    identical calls take the same route and return the same bits."""
    def volume(i, lens):            # voxels of input i when its deformed axes have extents `lens`
        v = float(numpy.prod([int(d) for d in shapes[i]], dtype=numpy.float64))
        for a, l in zip(plan.axis[i], lens):
            v *= float(l) / float(shapes[i][a])
        return v

    def at_least(i):                # no window is smaller than the output box plus margins and taps
        m = _WINDOW_MARGIN[names[i]][int(plan.order[i])]
        return [min(n_in, max(n_out + 2 * m + int(plan.order[i]) + 3, _WINDOW_MIN_LINE))
                for n_in, n_out in zip(in_len, out_len)]
    if not todo or _strict_crop:
        return False
    if disp_shape is not None and (len(disp_shape) < 2 or len(disp_shape) > 5 or
                                   numpy.prod([int(d) for d in disp_shape]) > 7680):
        return False                # (control grid in LDS, up to 4 deformed axes: edhip_source_window)
    # inputs the window kernels take: contiguous, every deformed axis at least a tile kernel's shortest line
    todo = [i for i in todo if (contiguous is None or contiguous[i]) and
            all(int(shapes[i][a]) >= _WINDOW_MIN_LINE for a in plan.axis[i])]
    if not todo:
        return False
    full = sum(volume(i, in_len) for i in todo)
    least = sum(volume(i, at_least(i)) for i in todo)
    return not (full - least < CROP_WINDOW_MIN_SAVING or least > CROP_WINDOW_MAX_FRACTION * full)


def _crop_windows(plan, xs, disp_desc, dflag, crop, prefilter, device, stream, grid_stays=False):
    """Per input: a device tensor holding its filter window (2 ints per dimension, edhip_source_window), or
    None for 'filter the whole array'.  Floating-point volumes of orders 2 / 3 outside 'exact' arithmetic: the
    window's coefficients equal the whole-volume ones to below the data's rounding, not bit for bit."""
    n = len(xs)
    wins = [None] * n
    if crop is None or not prefilter or (_flags & _lib.FLAG_EXACT) or _strict_crop:
        return wins
    todo = [i for i in range(n) if int(plan.order[i]) in _WINDOW_MARGIN.get(_dtype_name(xs[i]), {})]
    if not todo:
        return wins
    ax0 = plan.axis[0]
    in_len = [int(xs[0].shape[a]) for a in ax0]
    out_len = [int(plan.output_shapes[0][a]) for a in ax0]
    # (one rule -- grid size, layout, shortest line, pay-off -- for this path and for the repeat-call lanes)
    if not _crop_window_pays(plan, [tuple(int(d) for d in x.shape) for x in xs], [_dtype_name(x) for x in xs], todo,
                             in_len, out_len, disp_shape=list(disp_desc.shape)[:disp_desc.ndim],
                             contiguous=[bool(x.is_contiguous()) for x in xs]):
        return wins
    torch = _torch()
    for i in todo:
        x = xs[i]
        if not x.is_contiguous() or any(int(x.shape[a]) < _WINDOW_MIN_LINE for a in plan.axis[i]):
            continue
        name = _dtype_name(x)
        vn = 4 if name == 'float32' else 2
        win = torch.empty(2 * x.dim(), dtype=torch.int32, device=device)
        st = _lib.source_window(disp_desc, in_len, out_len, plan.output_offset, plan.inverse_affine,
                                tuple(int(v) for v in x.shape), plan.axis[i], int(plan.order[i]), int(plan.mode[i]),
                                _WINDOW_MARGIN[name][int(plan.order[i])], vn if int(x.shape[-1]) % vn == 0 else 1,
                                _WINDOW_MIN_LINE,
                                _flags | dflag | _lib.FLAG_FAST | (_lib.FLAG_GRID_STAYS if (grid_stays and dflag) else 0),
                                stream, win.data_ptr())
        if st == 0:
            wins[i] = win
            grid_stays = True       # (the next input's window: same grid, just filtered by THIS call)
    return wins


def _prefilter_displacement(displacement, device):
    """Order-3 prefilter of the control grid along every grid axis (deform_grid.py:166-169,
    :269-272); the output keeps the displacement's dtype like numpy.zeros_like there.

    Returns (grid, extra_flags): small grids (the normal case) are handed over raw and filtered
    inside edhip_deform in a single launch (EDHIP_FLAG_RAW_DISPLACEMENT, same arithmetic and the
    same per-axis rounding); larger ones go through edhip_spline_filter1d axis by axis."""
    if displacement.ndim < 2:
        return displacement, 0
    if displacement.numel() <= _lib.RAW_DISPLACEMENT_MAX_POINTS:
        return displacement, _lib.FLAG_RAW_DISPLACEMENT
    return _filter_axes(displacement, range(1, displacement.ndim), 3, False, device), 0


def _lane_lookup(gradient, X, displacement, order, mode, cval, crop, prefilter, axis, X_shape,
                 affine, rotate, zoom):
    """(signature, lane) of a repeat call on device tensors (_fastlane.py); (None, None) for every
    call the lane does not serve; (sig, None) for a signature seen for the first time."""
    if _reduced or affine is not None or rotate is not None or zoom is not None or not _fastlane.enabled:
        return None, None
    if _strict_crop or (gradient and _grad_accumulation != 'fixed'):
        return None, None           # (the lanes are built for the default routes)
    sig = _fastlane.signature(gradient, X, displacement, order, mode, cval, crop, prefilter, axis,
                              X_shape, _flags | _route_flags())
    if sig is None:
        return None, None
    lane = _fastlane.lookup(sig)
    if lane is None:
        return sig, None
    if lane is False or lane.window_pays:
        return None, None
    return sig, lane


def _lane_build(sig, gradient, xs, dd, plan, prefilter, X_shape, crop):
    """After the general path has served `sig` once: prepare the repeat-call lane for it."""
    if dd.ndim < 2 or dd.numel() > _lib.RAW_DISPLACEMENT_MAX_POINTS:
        _fastlane.remember(sig, False)
        return
    try:
        lane = _fastlane.Lane(_this, gradient, xs, dd, plan, prefilter, X_shape, _flags | _route_flags(), crop)
    except Exception as exc:     # noqa: BLE001 -- whatever went wrong, the computed result must not be lost
        # the result of this call is already computed: a lane that cannot be built is no lane -- but say so once
        warnings.warn("elasticdeform_amd: no repeat-call lane for this signature (%s: %s)" % (type(exc).__name__, exc),
                      RuntimeWarning, stacklevel=3)
        lane = False
    _fastlane.remember(sig, lane)


# Layouts whose deformed axes are not the innermost ones (channel-last volumes: X is (D, H, W, C) with
# axis=(0, 1, 2)) miss every kernel that wants unit stride along the last deformed axis -- the LDS-DMA
# staging of the tile kernels, the integer fast path -- and run on the general kernels with one strided
# load per sample (96^3 x 4 float32, order 3: 0.39 ms against 0.15 ms channel-first).  Such inputs are
# brought to "step axes first" on the device (one transpose each way, ~4 passes over the array at HBM
# speed) and go through the same public function with the trailing axes as the deformed ones.  Values
# are those of the channel-first call on the same data.
RELAYOUT_MIN_ELEMENTS = 1 << 16


def _relayout_perms(plan, Xs):
    """Per input: the permutation that moves its non-deformed axes to the front, or None when the
    deformed axes already are the trailing ones (or the array is small).  None when no input needs one."""
    perms = []
    for x, ax, o in zip(Xs, plan.axis, plan.order):
        nd = len(x.shape)
        # (order 0 is one strided load per voxel either way: the two transposes would cost more than they save)
        if tuple(ax) == tuple(range(nd - len(ax), nd)) or int(o) < 1 or \
                int(numpy.prod(x.shape)) < RELAYOUT_MIN_ELEMENTS:
            perms.append(None)
        else:
            perms.append([a for a in range(nd) if a not in ax] + list(ax))
    return perms if any(p is not None for p in perms) else None


def _inverse_perm(p):
    inv = [0] * len(p)
    for i, a in enumerate(p):
        inv[a] = i
    return inv


def deform_random_grid(X, sigma=25, points=3, order=3, mode='constant', cval=0.0,
                       crop=None, prefilter=True, axis=None,
                       affine=None, rotate=None, zoom=None):
    """
    Elastic deformation with a random square deformation grid (deform_grid.py:6-49): draws
    ``numpy.random.randn(ndim, *points) * sigma`` from NumPy's global RNG, exactly like the
    reference, and calls :func:`deform_grid`.
    """
    Xs = _host.normalize_inputs(X)
    axis, deform_shape = _host.normalize_axis_list(axis, Xs)
    if not isinstance(points, (list, tuple)):
        points = [points] * len(deform_shape)
    displacement = numpy.random.randn(len(deform_shape), *points) * sigma
    with _random_grid_hint(sigma, points, deform_shape):
        return deform_grid(X, displacement, order, mode, cval, crop, prefilter, axis,
                           affine, rotate, zoom)


def deform_grid(X, displacement, order=3, mode='constant', cval=0.0, crop=None, prefilter=True,
                axis=None, affine=None, rotate=None, zoom=None):
    """
    Elastic deformation with a deformation grid (deform_grid.py:52-179).

    X : array or list of arrays (numpy.ndarray or torch.Tensor); displacement : array of shape
    (naxis, n_0, ..., n_{naxis-1}); order 0..5, mode in {nearest, wrap, reflect, mirror,
    constant}, cval, crop (slices over the deformed axes), prefilter, axis, affine
    (naxis x naxis+1), rotate / zoom (2-D only) -- all with the reference's meaning, and order /
    mode / cval / axis may be per-input lists.  Returns the deformed array, or a list for a list.
    """
    sig, lane = _lane_lookup(False, X, displacement, order, mode, cval, crop, prefilter, axis, None,
                             affine, rotate, zoom)
    if lane is not None:
        return lane.run(_this, X, X if type(X) is list else (X,), displacement)
    Xs = _host.normalize_inputs(X)
    plan = _host.cached_plan(Xs, displacement, order, mode, cval, crop, axis, affine, rotate, zoom)

    torch = _torch()
    if _host.degenerate_axis([x.shape for x in Xs], plan.axis):
        # a deformed axis of length 1: every voxel maps to the constant, as in the reference (_host.degenerate_axis)
        res = [_constant_result(x, plan.output_shapes[i], plan.cval[i]) for i, x in enumerate(Xs)]
        return res if isinstance(X, list) else res[0]
    device = _device_for(list(Xs) + [displacement])
    perms = _relayout_perms(plan, Xs)
    if perms is not None:
        # (every argument has passed the reference's checks above, with the caller's own axes)
        with torch.cuda.device(device):
            Xp = [_to_device(x, device) for x in Xs]
            Xp = [x.permute(p).contiguous() if p is not None else x for x, p in zip(Xp, perms)]
            axis_p = [tuple(range(x.dim() - plan.naxis, x.dim())) if p is not None else tuple(ax)
                      for x, p, ax in zip(Xp, perms, plan.axis)]
            outs = deform_grid(Xp, _to_device(displacement, device), order, mode, cval, crop, prefilter, axis_p,
                               affine, rotate, zoom)
            outs = [o.permute(_inverse_perm(p)).contiguous() if p is not None else o for o, p in zip(outs, perms)]
            res = [_from_device(o, x) for o, x in zip(outs, Xs)]
        return res if isinstance(X, list) else res[0]
    with torch.cuda.device(device):
        stream = _stream(device)
        Xs_dev = [_to_device(x, device) for x in Xs]
        # reduced-precision opt-in: 16-bit float volumes (and integer images with order > 1) are
        # computed in float32 and narrowed at the end
        direct = [_direct16(x, plan.axis[i], int(plan.order[i]), prefilter, crop) for i, x in enumerate(Xs_dev)]
        wide = [None if direct[i] else _widen(x, int(plan.order[i]), prefilter) for i, x in enumerate(Xs_dev)]
        Xd = [w if w is not None else x for w, x in zip(wide, Xs_dev)]
        dd = _to_device(displacement, device)

        # the displacement is always prefiltered (deform_grid.py:166-169); the inputs along their
        # deformed axes (deform_grid.py:155-164): the whole array, or -- with a crop -- only the
        # window of it that the cropped output can reach
        df, dflag = _prefilter_displacement(dd, device)
        wins = _crop_windows(plan, Xd, _desc(df), dflag, crop, prefilter, device, stream)
        Xf, in_descs = [], []
        for i, x in enumerate(Xd):
            xf = None
            if direct[i]:
                # 16-bit storage: the first filter pass widens, the chain continues in place in float32
                xf = torch.empty(x.shape, dtype=torch.float32, device=device)
                if _lib.spline_filter_axes(_desc(x), _desc(xf), list(plan.axis[i]), int(plan.order[i]), False,
                                           _flags | _lib.FLAG_FAST, stream, may_decline=True) != 0:
                    direct[i] = False
                    wide[i] = x = Xd[i] = x.to(torch.float32)
                    xf = None
            if xf is not None:
                pass
            elif not (prefilter and plan.order[i] > 1):
                xf = x
            elif wins[i] is not None:
                # a full-size buffer of which only the window is written -- and read: the kernels' taps stay
                # inside the source box, the box inside the window
                xf = torch.empty_like(x)
                if _lib.spline_filter_axes_window(_desc(x), _desc(xf), list(plan.axis[i]), int(plan.order[i]), False,
                                                  wins[i].data_ptr(), _flags, stream) != 0:
                    xf = None           # (outside the tile kernels' envelope: nothing was launched)
            if xf is None:
                xf = _filter_axes(x, plan.axis[i], int(plan.order[i]), False, device, stream=stream)
            in_descs.append(_desc(xf))
            Xf.append(xf)                  # keeps the buffers alive until the launch is enqueued

        # every output element is written by the kernel (value or cval), so no zero fill is needed
        # (a 16-bit volume that stayed in 16 bits: K1 reads the float32 coefficients and narrows at its store)
        outs = [torch.empty(tuple(int(s) for s in shape), dtype=x.dtype, device=device)
                for shape, x in zip(plan.output_shapes, Xd)]

        bflag = _box_flag_forward(displacement, df, device, stream)
        if dflag and any(w is not None for w in wins):
            bflag |= _lib.FLAG_GRID_STAYS        # (edhip_source_window has just filtered this very grid on this stream)
        fast16 = _lib.FLAG_FAST if any(direct) else 0
        if _lib.deform(False, in_descs, _desc(df), plan.output_offset,
                       [_desc(o) for o in outs], plan.axis, plan.order, plan.mode, plan.cval,
                       plan.inverse_affine, _flags | dflag | bflag | fast16 | _route_flags(), stream, prepared=_prepared(plan, len(Xd)),
                       may_decline=bool(fast16)) != 0:
            # the library declined the 16-bit stores: float32 outputs, narrowed by a cast like the other route
            outs = [torch.empty(o.shape, dtype=torch.float32, device=device) if d else o for o, d in zip(outs, direct)]
            wide = [xf if d else w for xf, d, w in zip(Xf, direct, wide)]
            direct = [False] * len(direct)
            _lib.deform(False, in_descs, _desc(df), plan.output_offset,
                        [_desc(o) for o in outs], plan.axis, plan.order, plan.mode, plan.cval,
                        plan.inverse_affine, _flags | dflag | bflag | _route_flags(), stream, prepared=_prepared(plan, len(Xd)))
        outs = [_narrow(o, xs) if w is not None else o for o, xs, w in zip(outs, Xs_dev, wide)]
        res = [_from_device(o, x) for o, x in zip(outs, Xs)]
        if sig is not None:
            _lane_build(sig, False, Xs_dev, dd, plan, prefilter, None, crop)
    return res if isinstance(X, list) else res[0]


def deform_grid_gradient(dY, displacement, order=3, mode='constant', cval=0.0, crop=None,
                         prefilter=True, axis=None, X_shape=None,
                         affine=None, rotate=None, zoom=None):
    """
    Gradient of :func:`deform_grid` with respect to its input (deform_grid.py:182-291): the exact
    adjoint, interpolation and prefilter included.  ``X_shape`` (tuple, or list of tuples) is
    required when ``crop`` is used.
    """
    sig, lane = _lane_lookup(True, dY, displacement, order, mode, cval, crop, prefilter, axis, X_shape,
                             affine, rotate, zoom)
    if lane is not None:
        return lane.run(_this, dY, dY if type(dY) is list else (dY,), displacement)
    dYs = _host.normalize_inputs(dY)

    if isinstance(X_shape, tuple):
        X_shape = [X_shape]
    elif X_shape is None:
        if crop is not None:
            raise ValueError("X_shape is required if the crop parameter is given.")
        X_shape = [tuple(dy.shape) for dy in dYs]

    # every argument check runs before anything touches the device, in the reference's order
    # (deform_grid.py:246-266): a bad displacement / order / crop raises what the reference raises
    plan = _host.cached_plan([_host.ShapeOnly(s) for s in X_shape], displacement, order, mode, cval, crop,
                             axis, affine, rotate, zoom)
    if [tuple(s) for s in plan.output_shapes] != [tuple(dy.shape) for dy in dYs]:
        raise ValueError("X_shape does not match output shape and cropping. "
                         "Expected output shape is %s, but %s given."
                         % (str(plan.output_shapes), str([tuple(dy.shape) for dy in dYs])))

    torch = _torch()
    if _host.degenerate_axis(X_shape, plan.axis):
        # (a deformed axis of length 1: no voxel contributes, the gradient is zero -- deform.c:928)
        res = [_constant_result(dy, tuple(int(v) for v in sh), 0.0) for dy, sh in zip(dYs, X_shape)]
        return res if isinstance(dY, list) else res[0]
    device = _device_for(list(dYs) + [displacement])
    perms = _relayout_perms(plan, [_host.ShapeOnly(sh) for sh in X_shape])
    if perms is not None:
        with torch.cuda.device(device):
            dYp = [_to_device(dy, device) for dy in dYs]
            dYp = [dy.permute(p).contiguous() if p is not None else dy for dy, p in zip(dYp, perms)]
            axis_p = [tuple(range(dy.dim() - plan.naxis, dy.dim())) if p is not None else tuple(ax)
                      for dy, p, ax in zip(dYp, perms, plan.axis)]
            shape_p = [tuple(int(sh[a]) for a in p) if p is not None else tuple(sh) for sh, p in zip(X_shape, perms)]
            dXs = deform_grid_gradient(dYp, _to_device(displacement, device), order, mode, cval, crop, prefilter,
                                       axis_p, shape_p, affine, rotate, zoom)
            dXs = [g.permute(_inverse_perm(p)).contiguous() if p is not None else g for g, p in zip(dXs, perms)]
            res = [_from_device(g, dy) for g, dy in zip(dXs, dYs)]
        return res if isinstance(dY, list) else res[0]
    with torch.cuda.device(device):
        dY_dev = [_to_device(dy, device) for dy in dYs]
        # 16-bit dY that can stay in 16 bits (_direct16): K2 widens it where it reads it, the accumulators are float32
        direct = [_direct16(dy, plan.axis[i], int(plan.order[i]), prefilter, crop) and tuple(X_shape[i]) == tuple(dy.shape)
                  for i, dy in enumerate(dY_dev)]
        wide = [None if direct[i] else _widen(dy, int(plan.order[i]), prefilter) for i, dy in enumerate(dY_dev)]
        dYd = [w if w is not None else dy for w, dy in zip(wide, dY_dev)]
        # gradient accumulators start at zero (deform_grid.py:243): cleared by the library next to its
        # tables kernel (EDHIP_FLAG_ZERO_GRADIENT) instead of by a fill launch of their own
        dXs = [torch.empty(tuple(int(v) for v in s), dtype=torch.float32 if d else dy.dtype, device=device)
               for s, dy, d in zip(X_shape, dYd, direct)]

        dd = _to_device(displacement, device)
        df, dflag = _prefilter_displacement(dd, device)

        stream = _stream(device)
        gflags = _flags | dflag | _lib.FLAG_ZERO_GRADIENT | _box_flag_gradient(displacement, df, device, stream)
        if _grad_accumulation == 'float':
            # floating-point atomics in the array's own type (set_gradient_accumulation): the exact kernel's scatter
            gflags = (gflags & ~(_lib.FLAG_FAST | _lib.FLAG_AUTO)) | _lib.FLAG_EXACT
        if _lib.deform(True, [_desc(x) for x in dXs], _desc(df), plan.output_offset,
                       [_desc(dy) for dy in dYd], plan.axis, plan.order, plan.mode, plan.cval,
                       plan.inverse_affine, gflags | (_lib.FLAG_FAST if any(direct) else 0),
                       stream, prepared=_prepared(plan, len(dXs)), may_decline=any(direct)) != 0:
            # the library declined the 16-bit loads: widen dY with a cast, like the other route
            wide = [dy.to(torch.float32) if d else w for dy, d, w in zip(dY_dev, direct, wide)]
            dYd = [w if w is not None else dy for w, dy in zip(wide, dY_dev)]
            direct = [False] * len(direct)
            _lib.deform(True, [_desc(x) for x in dXs], _desc(df), plan.output_offset,
                        [_desc(dy) for dy in dYd], plan.axis, plan.order, plan.mode, plan.cval,
                        plan.inverse_affine, gflags, stream, prepared=_prepared(plan, len(dXs)))

        # gradient of the prefilter: its transpose along each deformed axis (deform_grid.py:276-286).
        # With a crop the scatter only touched a box of dX: the transposed filter runs on that box
        # plus its decay margin and the result replaces the box (the rest stays exactly zero, where
        # the whole-volume filter would leave values below 1e-18 of the gradient's scale).
        wins = _crop_windows(plan, dXs, _desc(df), dflag, crop, prefilter, device, stream, grid_stays=True)
        dXf = []
        for i, x in enumerate(dXs):
            if direct[i]:
                # the transposed chain in place in float32; its last pass narrows into the 16-bit result
                g16 = torch.empty(x.shape, dtype=dY_dev[i].dtype, device=device)
                if _lib.spline_filter_axes(_desc(x), _desc(g16), list(plan.axis[i]), int(plan.order[i]), True,
                                           _flags | _lib.FLAG_FAST | _lib.FLAG_SCRATCH_INPUT, stream,
                                           may_decline=True) == 0:
                    dXf.append(g16)
                    continue
                direct[i] = False
                wide[i] = x                 # (float32 chain below, narrowed by a cast)
            if not (prefilter and plan.order[i] > 1):
                dXf.append(x)
            elif wins[i] is not None and _lib.spline_filter_axes_window(
                    _desc(x), _desc(x), list(plan.axis[i]), int(plan.order[i]), True, wins[i].data_ptr(), _flags,
                    stream) == 0:
                dXf.append(x)           # in place, inside the window; the rest of dX stays exactly zero
            else:
                dXf.append(_filter_axes(x, plan.axis[i], int(plan.order[i]), True, device, overwrite=True,
                                        stream=stream))
        dXf = [_narrow(x, dy) if w is not None else x for x, dy, w in zip(dXf, dY_dev, wide)]
        res = [_from_device(x, dy) for x, dy in zip(dXf, dYs)]
        if sig is not None:
            _lane_build(sig, True, dY_dev, dd, plan, prefilter, X_shape, crop)
    return res if isinstance(dY, list) else res[0]


# ---- batches: one control grid per sample -----------------------

def _batch_plan(X, displacements, order, mode, cval, crop, axis, affine, rotate, zoom):
    """Normalise a batched call: X is (B, ...) -- sample b is X[b] -- and displacements is
    (B, naxis, n_0, ...).  `axis` counts the axes of ONE sample (like deform_grid's).  Returns the
    Plan of a single sample (shared by the whole batch)."""
    if not _host.is_array(X) or X.ndim < 2:
        raise Exception('X should be an array with a leading batch axis.')
    if not _host.is_array(displacements) or displacements.ndim < 3:
        raise Exception('displacements should be an array of shape (batch, naxis, n_0, ...).')
    assert displacements.shape[0] == X.shape[0], 'One displacement grid per sample is required.'
    assert not isinstance(order, (list, tuple)) and not isinstance(mode, (list, tuple)) and \
        not isinstance(cval, (list, tuple)), 'order, mode and cval are shared by the batch.'
    return _host.Plan([X[0]], displacements[0], order, mode, cval, crop, axis, affine, rotate, zoom)


def deform_grid_batch(X, displacements, order=3, mode='constant', cval=0.0, crop=None,
                      prefilter=True, axis=None, affine=None, rotate=None, zoom=None):
    """
    :func:`deform_grid` over a batch with ONE CONTROL GRID PER SAMPLE: ``X`` has shape
    ``(B, ...)``, ``displacements`` ``(B, naxis, n_0, ..., n_{naxis-1})``; every other argument
    has the meaning it has for a single sample and is shared.  Returns ``(B, ...)``.

    Equivalent to ``stack([deform_grid(X[b], displacements[b], ...) for b in range(B)])`` --
    same results, bit for bit -- but the B samples are prefiltered together (the batch axis is just
    another outer axis of the filter passes) and deformed by ONE set of kernel launches
    (``edhip_deform_batch_strided``: the strip index of the tile kernels carries the sample), which
    removes the per-sample launches and host overhead that dominate for small volumes.  The reference has no batched entry point (one grid per call, deform_grid.py:52).
    """
    plan = _batch_plan(X, displacements, order, mode, cval, crop, axis, affine, rotate, zoom)
    torch = _torch()
    device = _device_for([X, displacements])
    with torch.cuda.device(device):
        Xd = _to_device(X, device)
        dd = _to_device(displacements, device)
        B = int(Xd.shape[0])
        ax = plan.axis[0]
        o = int(plan.order[0])
        Xf = Xd
        if prefilter and o > 1:
            Xf = _filter_axes(Xd, [a + 1 for a in ax], o, False, device)
        # all B control grids are prefiltered together (three launches for the batch; same values as
        # the per-call RAW_DISPLACEMENT path: a test pins that), then ONE library call and -- for
        # float volumes with 3 deformed axes -- one tables launch + one tile launch for all B samples
        df = _filter_axes(dd, range(2, dd.ndim), 3, False, device)
        out = torch.empty((B,) + tuple(int(v) for v in plan.output_shapes[0]), dtype=Xd.dtype, device=device)
        (xd, xs), (dd0, ds), (od, os_) = _desc_sample0(Xf), _desc_sample0(df), _desc_sample0(out)
        # the buffer of the filtered grids is kept for the gradient call of the same displacement
        # tensor: it takes this call's tile boxes (see _box_owner and deform_grid_gradient_batch)
        stream = _stream(device)
        ident = _box_id(displacements, dd)
        _batch_grids[(device.index, stream)] = (ident, df) if ident is not None else None
        _box_owner[(device.index, stream)] = None
        _lib.deform_batch_strided(False, B, xd, xs, dd0, ds, plan.output_offset, od, os_, ax, o,
                                  int(plan.mode[0]), float(plan.cval[0]), plan.inverse_affine,
                                  _flags | _route_flags() | (_lib.FLAG_KEEP_BOXES if ident is not None else 0), stream)
        return _from_device(out, X)


def deform_grid_gradient_batch(dY, displacements, order=3, mode='constant', cval=0.0, crop=None,
                               prefilter=True, axis=None, X_shape=None, affine=None, rotate=None,
                               zoom=None):
    """Gradient of :func:`deform_grid_batch` with respect to ``X``.  ``X_shape`` is the shape of
    ONE sample (required with a crop)."""
    if not _host.is_array(dY) or dY.ndim < 2:
        raise Exception('dY should be an array with a leading batch axis.')
    if X_shape is None:
        if crop is not None:
            raise ValueError("X_shape is required if the crop parameter is given.")
        X_shape = tuple(dY.shape[1:])
    if not _host.is_array(displacements) or displacements.ndim < 3:
        raise Exception('displacements should be an array of shape (batch, naxis, n_0, ...).')
    assert displacements.shape[0] == dY.shape[0], 'One displacement grid per sample is required.'
    assert not isinstance(order, (list, tuple)) and not isinstance(mode, (list, tuple)) and \
        not isinstance(cval, (list, tuple)), 'order, mode and cval are shared by the batch.'
    plan = _host.Plan([_host.ShapeOnly(X_shape)], displacements[0], order, mode, cval, crop, axis,
                      affine, rotate, zoom)
    if tuple(plan.output_shapes[0]) != tuple(dY.shape[1:]):
        raise ValueError("X_shape does not match output shape and cropping. "
                         "Expected output shape is %s, but %s given."
                         % (str(plan.output_shapes[0]), str(tuple(dY.shape[1:]))))
    torch = _torch()
    device = _device_for([dY, displacements])
    with torch.cuda.device(device):
        dYd = _to_device(dY, device)
        dd = _to_device(displacements, device)
        B = int(dYd.shape[0])
        dX = torch.zeros((B,) + tuple(int(v) for v in X_shape), dtype=dYd.dtype, device=device)
        ax = plan.axis[0]
        o = int(plan.order[0])
        stream = _stream(device)
        kept = _batch_grids.get((device.index, stream))
        ident = _box_id(displacements, dd)
        bflag = 0
        # The control grids are ALWAYS prefiltered again (three small launches): storage address +
        # version counter say nothing certain about contents (a new tensor on a freed one's address, a
        # `.data` write).  Only the forward call's tile boxes are handed over -- the kernel treats
        # them as a hint it verifies per voxel -- and for that the fresh grids are written into the
        # forward call's buffer, whose address is the library's key for the boxes.
        df = _filter_axes(dd, range(2, dd.ndim), 3, False, device)
        if kept is not None and ident is not None and kept[0] == ident and kept[1].shape == df.shape \
                and kept[1].dtype == df.dtype:
            kept[1].copy_(df)
            df = kept[1]
            bflag = _lib.FLAG_USE_BOXES
        (xd, xs), (dd0, ds), (yd, ys) = _desc_sample0(dX), _desc_sample0(df), _desc_sample0(dYd)
        _lib.deform_batch_strided(True, B, xd, xs, dd0, ds, plan.output_offset, yd, ys, ax, o,
                                  int(plan.mode[0]), float(plan.cval[0]), plan.inverse_affine, _flags | bflag,
                                  stream)
        if prefilter and o > 1:
            dX = _filter_axes(dX, [a + 1 for a in ax], o, True, device, overwrite=True)
        return _from_device(dX, dY)
