"""
Repeat-call lane of ``deform_grid`` / ``deform_grid_gradient`` for tensors that already live in HBM.

Augmentation pipelines call the same function thousands of times with new data of the same layout.
For a 32^3 patch the kernels take a few microseconds and the call is bound by the host: argument
normalisation, descriptor construction, per-call ctypes conversions (`profiles/r03_host_latency.txt`).
This module memoises everything of a call that does not depend on the data -- the validated Plan,
the ctypes descriptors of every array, the parameter arrays, the filter schedule -- keyed on the
layout signature of the arguments; a repeat call allocates its outputs, patches the data pointers
into the prepared descriptors and goes straight to the C ABI (include/edhip.h).

It is a cache of the general path in deform_grid.py, not a second implementation: a lane is only
built after the general path has served the same signature once (so every argument check of the
reference, deform_grid.py:52-179 / :182-291, has passed for it), it issues the same library calls
with the same arguments, and anything it does not recognise -- NumPy arrays, arrays on another
device, affine / rotate / zoom, the reduced-precision opt-in, control grids too large for the
in-library prefilter, crops large enough for the crop-aware prefilter -- is left to the general path.
"""
import ctypes
import threading

from . import _host
from . import _lib

enabled = True          # tests and tools/latency_small.py switch the lane off to compare with the general path
_MAX_LANES = 256
_lanes = {}
_torch = None
_raw_stream = None


def _init():
    global _torch, _raw_stream
    import torch
    _torch = torch
    _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if _raw_stream is None:
        _raw_stream = lambda idx: torch.cuda.current_stream(idx).cuda_stream      # noqa: E731


def _hashable(v):
    """order / mode / cval / axis: tagged by type like the general path's plan key (_host._freeze), so that
    order=3, order=3.0 and order=True -- which the general path validates differently -- never share a lane"""
    return None if v is None else _host._freeze(v)


class _NoLane(Exception):
    pass


def _crop_key(crop):
    if crop is None:
        return None
    if not all(isinstance(s, slice) for s in crop):
        raise _NoLane()             # (the general path raises the reference's exception for it)
    return tuple((s.start, s.stop, s.step) for s in crop)


def signature(gradient, X, displacement, order, mode, cval, crop, prefilter, axis, X_shape, flags):
    """Layout signature of a call, or None when the call is not one this lane serves."""
    if _torch is None:
        _init()
    torch = _torch
    if type(X) is list:
        xs = X
        if not xs:
            return None
    elif isinstance(X, torch.Tensor):
        xs = (X,)
    else:
        return None
    if not isinstance(displacement, torch.Tensor) or not displacement.is_cuda:
        return None
    dev = displacement.device.index
    if dev != torch.cuda.current_device():
        return None
    sig = []
    for x in xs:
        if not isinstance(x, torch.Tensor) or not x.is_cuda or x.device.index != dev:
            return None
        sig.append((x.shape, x.stride(), x.dtype))
    try:
        if X_shape is not None:
            X_shape = tuple(X_shape) if isinstance(X_shape, tuple) else tuple(tuple(s) for s in X_shape)
        return (gradient, type(X) is list, tuple(sig), dev, displacement.shape, displacement.stride(),
                displacement.dtype, _hashable(order), _hashable(mode), _hashable(cval), _crop_key(crop),
                bool(prefilter), _hashable(axis), X_shape, flags)
    except (_NoLane, TypeError):
        return None


def _like_desc(dgm, x, shape=None):
    """Descriptor (data pointer 0) of what torch.empty_like(x) / torch.empty(shape, dtype=x.dtype) would be for
    the dense tensors the lanes serve -- built from the layout, no allocation."""
    es = x.element_size()
    if shape is None:
        if not x.is_contiguous():
            return dgm._desc(_torch.empty_like(x))          # (preserved strides of a non-dense view: ask torch)
        shape = x.shape
    shape = tuple(int(v) for v in shape)
    strides, st = [], es
    for d in reversed(shape):
        strides.append(st)
        st *= d
    return _lib.describe(0, dgm._dtype_name(x), shape, list(reversed(strides)))


class _Filter(object):
    """One prefilter chain (deform_grid.py:157-162 / :279-284) with its descriptors prepared: a single
    edhip_spline_filter_axes call when every line fits the whole-line kernels, else one
    edhip_spline_filter1d call per axis, ping-ponging between two temporaries."""
    __slots__ = ("chain", "steps", "src", "b0", "b1", "axes_c", "n", "order", "transpose", "overwrite")

    def __init__(self, dgm, x, axes, order, transpose, overwrite):
        axes = [int(a) for a in axes]
        self.order = int(order)
        self.transpose = 1 if transpose else 0
        self.n = len(axes)
        self.axes_c = (ctypes.c_int32 * self.n)(*axes)
        self.chain = all(int(x.shape[d]) <= 256 for d in axes)
        self.overwrite = bool(overwrite and self.chain)
        self.src = dgm._desc(x)
        # (descriptors of tensors "like x" from its layout alone: only the data pointer changes per call; building
        # them from scratch tensors cost up to three volume-sized allocations after the result was already computed)
        self.b0 = self.src if self.overwrite else _like_desc(dgm, x)
        self.b1 = _like_desc(dgm, x) if (self.n > 1 and not self.chain) else None
        self.steps = axes

    def run(self, L, x, flags, stream, ebuf):
        """-> (filtered tensor, status)"""
        self.src.data = x.data_ptr()
        if self.chain:
            dst = x if self.overwrite else _torch.empty_like(x)
            if not self.overwrite:
                self.b0.data = dst.data_ptr()
            st = L.edhip_spline_filter_axes(ctypes.byref(self.src), ctypes.byref(self.b0), self.n, self.axes_c,
                                            self.order, self.transpose, flags, stream, ebuf, 256)
            return dst, st
        t0 = _torch.empty_like(x)
        self.b0.data = t0.data_ptr()
        t1 = None
        if self.b1 is not None:
            t1 = _torch.empty_like(x)
            self.b1.data = t1.data_ptr()
        src, cur = self.src, x
        for i, d in enumerate(self.steps):
            dst, dt = (self.b0, t0) if not (i & 1) else (self.b1, t1)
            st = L.edhip_spline_filter1d(ctypes.byref(src), ctypes.byref(dst), d, self.order, self.transpose,
                                         flags, stream, ebuf, 256)
            if st:
                return None, st
            src, cur = dst, dt
        return cur, 0


class Lane(object):
    """Everything data-independent of one deform_grid / deform_grid_gradient signature."""

    def __init__(self, dgm, gradient, xs, displacement, plan, prefilter, X_shape, flags, crop):
        torch = _torch
        n = len(xs)
        self.gradient = gradient
        self.n = n
        self.lock = threading.Lock()
        self.args = dgm._prepared(plan, n)
        self.flags = int(flags) | _lib.FLAG_RAW_DISPLACEMENT
        self.disp = dgm._desc(displacement)
        self.dev = displacement.device
        self.ins = (_lib.EdhipArray * n)()
        self.outs = (_lib.EdhipArray * n)()
        self.filters = [None] * n
        if gradient:
            # inputs of the library call = dX (zero-filled accumulators), outputs = dY
            self.shapes = [tuple(int(v) for v in s) for s in X_shape]
            self.dtypes = [x.dtype for x in xs]
            for i, x in enumerate(xs):
                dx = torch.empty(self.shapes[i], dtype=x.dtype, device=self.dev)
                self.ins[i] = dgm._desc(dx)
                self.outs[i] = dgm._desc(x)
                if prefilter and plan.order[i] > 1:
                    self.filters[i] = _Filter(dgm, dx, plan.axis[i], plan.order[i], True, True)
        else:
            self.shapes = [tuple(int(v) for v in s) for s in plan.output_shapes]
            self.dtypes = [x.dtype for x in xs]
            for i, x in enumerate(xs):
                if prefilter and plan.order[i] > 1:
                    self.filters[i] = _Filter(dgm, x, plan.axis[i], plan.order[i], False, False)
                    self.ins[i] = _like_desc(dgm, x)
                else:
                    self.ins[i] = dgm._desc(x)
                self.outs[i] = _like_desc(dgm, x, self.shapes[i])
        # would the crop-aware prefilter engage (deform_grid._crop_windows)?  Such calls stay on the general path;
        # the rule is the general path's own (one helper), so that repeated identical calls take one route
        self.window_pays = False
        if crop is not None and prefilter and not (flags & _lib.FLAG_EXACT):
            in_shapes = self.shapes if gradient else [tuple(int(d) for d in x.shape) for x in xs]
            names = [str(dt).replace('torch.', '') for dt in self.dtypes]
            todo = [i for i in range(n) if int(plan.order[i]) in dgm._WINDOW_MARGIN.get(names[i], {})]
            ax0 = plan.axis[0]
            in_len = [int(in_shapes[0][a]) for a in ax0]
            out_len = [int(plan.output_shapes[0][a]) for a in ax0]
            # (gradient lanes filter dX, which the lane allocates contiguous; forward lanes filter the inputs as given)
            self.window_pays = dgm._crop_window_pays(plan, in_shapes, names, todo, in_len, out_len,
                                                     disp_shape=[int(d) for d in displacement.shape],
                                                     contiguous=[True] * n if gradient else [bool(x.is_contiguous()) for x in xs])

    def run(self, dgm, X, xs, displacement):
        torch = _torch
        L = _lib.load()
        a = self.args
        ebuf = _lib._buf()
        stream = _raw_stream(self.dev.index)
        keep = []
        with self.lock:
            self.disp.data = displacement.data_ptr()
            if self.gradient:
                res = []
                for i in range(self.n):
                    dx = torch.empty(self.shapes[i], dtype=self.dtypes[i], device=self.dev)      # cleared by the library
                    self.ins[i].data = dx.data_ptr()
                    self.outs[i].data = xs[i].data_ptr()
                    res.append(dx)
                bflag = dgm._box_flag_gradient(displacement, displacement, self.dev, stream)
                st = L.edhip_deform(1, self.n, self.ins, ctypes.byref(self.disp), a.off, self.outs, a.naxis,
                                    a.axis, a.orders, a.modes, a.cvals, a.aff,
                                    self.flags | bflag | _lib.FLAG_ZERO_GRADIENT, stream, ebuf, 256)
                if st:
                    _lib.raise_for_status(st, ebuf)
                for i, f in enumerate(self.filters):
                    if f is not None:
                        res[i], st = f.run(L, res[i], self.flags & ~(_lib.FLAG_RAW_DISPLACEMENT | _lib.FLAG_STRONG_FIELD), stream, ebuf)
                        if st:
                            _lib.raise_for_status(st, ebuf)
            else:
                res = []
                for i in range(self.n):
                    o = torch.empty(self.shapes[i], dtype=self.dtypes[i], device=self.dev)
                    self.outs[i].data = o.data_ptr()
                    res.append(o)
                bflag = dgm._box_flag_forward(displacement, displacement, self.dev, stream)
                for i, f in enumerate(self.filters):
                    x = xs[i]
                    if f is not None:
                        x, st = f.run(L, x, self.flags & ~(_lib.FLAG_RAW_DISPLACEMENT | _lib.FLAG_STRONG_FIELD), stream, ebuf)
                        if st:
                            _lib.raise_for_status(st, ebuf)
                        keep.append(x)
                    self.ins[i].data = x.data_ptr()
                st = L.edhip_deform(0, self.n, self.ins, ctypes.byref(self.disp), a.off, self.outs, a.naxis,
                                    a.axis, a.orders, a.modes, a.cvals, a.aff, self.flags | bflag, stream,
                                    ebuf, 256)
                if st:
                    _lib.raise_for_status(st, ebuf)
        return res if type(X) is list else res[0]


def lookup(sig):
    return _lanes.get(sig) if sig is not None else None


def remember(sig, lane):
    if len(_lanes) >= _MAX_LANES:
        _lanes.clear()
    _lanes[sig] = lane


def clear():
    _lanes.clear()
