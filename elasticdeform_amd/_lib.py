"""
ctypes binding of the C ABI (include/edhip.h) exported by elasticdeform_amd/libedhip.so -- the
hand-written HIP library for gfx950.  This is the Python side of the drop-in boundary: it plays
the role the CPython extension ``_deform_grid`` plays in the reference
(/root/reference/elasticdeform/_deform_grid.c:306-311).

There is NO CPU fallback: if the library is missing or no GPU is visible the product fails loudly.
"""
import ctypes
import os
import threading

import numpy

MAX_DIMS = 8
MAX_AXES = 7

FLAG_AUTO, FLAG_EXACT, FLAG_FAST = 0, 1, 2
FLAG_RAW_DISPLACEMENT = 4      # edhip_deform prefilters the control grid itself (<= 4096 points)
FLAG_GRID_STAYS = 64           # with RAW_DISPLACEMENT: the raw grid of the previous RAW call on the stream, unchanged
FLAG_SCRATCH_INPUT = 128       # edhip_spline_filter_axes: in place on the input, the last pass input -> output
FLAG_STRONG_FIELD = 256        # forward hint: a strong displacement field -> the z-walk route for every geometry it supports
ERR_UNSUPPORTED = 5             # EDHIP_ERR_UNSUPPORTED: legal, outside this build's limits (nothing was launched)
FLAG_KEEP_BOXES = 8            # forward: leave the tiles' bounding boxes for the gradient call that follows
FLAG_USE_BOXES = 16            # gradient: same displacement contents and geometry as that forward call
FLAG_ZERO_GRADIENT = 32        # gradient: the library clears the (dense) accumulators itself before scattering
RAW_DISPLACEMENT_MAX_POINTS = 4096

# enum edhip_dtype
DTYPE_CODES = {
    'bool': 0, 'uint8': 1, 'int8': 2, 'uint16': 3, 'int16': 4, 'uint32': 5, 'int32': 6,
    'uint64': 7, 'int64': 8, 'float32': 9, 'float64': 10,
    # reduced-precision storage: an extension the host layer only uses after an explicit opt-in
    # (the reference rejects half precision, deform.c:742-747)
    'float16': 11, 'bfloat16': 12,
}
REDUCED_DTYPES = ('float16', 'bfloat16')

# enum edhip_status -> the exception class the reference raises for that condition
_STATUS_EXC = {
    1: RuntimeError,   # EDHIP_ERR_INVALID      (PyErr_SetString(PyExc_RuntimeError, ...), _deform_grid.c:121-255)
    2: RuntimeError,   # EDHIP_ERR_DTYPE        ('data type not supported', deform.c:744,891,922)
    3: MemoryError,    # EDHIP_ERR_MEMORY       (PyErr_NoMemory, deform.c:394-398)
    4: RuntimeError,   # EDHIP_ERR_DEVICE
    5: RuntimeError,   # EDHIP_ERR_UNSUPPORTED
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libedhip.so')

# every symbol include/edhip.h declares
EXPORTS = ('edhip_version', 'edhip_status_string', 'edhip_device_count', 'edhip_deform',
           'edhip_deform_batch', 'edhip_deform_batch_strided', 'edhip_source_box', 'edhip_spline_filter1d',
           'edhip_spline_filter_axes', 'edhip_source_window', 'edhip_spline_filter_axes_window',
           'edhip_release_scratch', 'edhip_profile_dominant',
           'edhip_profile_last_us')


class EdhipArray(ctypes.Structure):
    """struct edhip_array"""
    _fields_ = [('data', ctypes.c_void_p), ('dtype', ctypes.c_int32), ('ndim', ctypes.c_int32),
                ('shape', ctypes.c_int64 * MAX_DIMS), ('stride_bytes', ctypes.c_int64 * MAX_DIMS)]


_lib = None
_lock = threading.Lock()


def load():
    """Load libedhip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'elasticdeform_amd: %s not found -- build the HIP extension first '
                '(`make -C elasticdeform_amd/csrc` or `python -c "import __graft_entry__ as g; '
                'g.build()"`).  There is no CPU fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.edhip_version.restype = ctypes.c_int
        L.edhip_version.argtypes = []
        L.edhip_status_string.restype = ctypes.c_char_p
        L.edhip_status_string.argtypes = [ctypes.c_int]
        L.edhip_device_count.restype = ctypes.c_int
        L.edhip_device_count.argtypes = []
        L.edhip_deform.restype = ctypes.c_int
        L.edhip_deform.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.POINTER(EdhipArray), ctypes.POINTER(EdhipArray),
            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(EdhipArray), ctypes.c_int,
            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double),
            ctypes.POINTER(ctypes.c_double), ctypes.c_uint32, ctypes.c_void_p, ctypes.c_char_p,
            ctypes.c_size_t]
        L.edhip_release_scratch.restype = ctypes.c_int
        L.edhip_release_scratch.argtypes = []
        L.edhip_deform_batch.restype = ctypes.c_int
        L.edhip_deform_batch.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.POINTER(EdhipArray), ctypes.POINTER(EdhipArray),
            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(EdhipArray), ctypes.c_int,
            ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, ctypes.c_double,
            ctypes.POINTER(ctypes.c_double), ctypes.c_uint32, ctypes.c_void_p, ctypes.c_char_p,
            ctypes.c_size_t]
        L.edhip_deform_batch_strided.restype = ctypes.c_int
        L.edhip_deform_batch_strided.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.POINTER(EdhipArray), ctypes.c_int64,
            ctypes.POINTER(EdhipArray), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
            ctypes.POINTER(EdhipArray), ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int32),
            ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
            ctypes.c_uint32, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        L.edhip_profile_dominant.restype = ctypes.c_int
        L.edhip_profile_dominant.argtypes = [ctypes.c_int]
        L.edhip_profile_last_us.restype = ctypes.c_double
        L.edhip_profile_last_us.argtypes = []
        L.edhip_source_box.restype = ctypes.c_int
        L.edhip_source_box.argtypes = [
            ctypes.POINTER(EdhipArray), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
            ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.POINTER(ctypes.c_double),
            ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p,
            ctypes.c_size_t]
        L.edhip_spline_filter1d.restype = ctypes.c_int
        L.edhip_spline_filter1d.argtypes = [
            ctypes.POINTER(EdhipArray), ctypes.POINTER(EdhipArray), ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        L.edhip_spline_filter_axes.restype = ctypes.c_int
        L.edhip_spline_filter_axes.argtypes = [
            ctypes.POINTER(EdhipArray), ctypes.POINTER(EdhipArray), ctypes.c_int,
            ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p,
            ctypes.c_char_p, ctypes.c_size_t]
        L.edhip_source_window.restype = ctypes.c_int
        L.edhip_source_window.argtypes = [
            ctypes.POINTER(EdhipArray), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
            ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int,
            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_char_p, ctypes.c_size_t]
        L.edhip_spline_filter_axes_window.restype = ctypes.c_int
        L.edhip_spline_filter_axes_window.argtypes = [
            ctypes.POINTER(EdhipArray), ctypes.POINTER(EdhipArray), ctypes.c_int,
            ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
            ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _lib = L
    return _lib


def raise_for_status(status, errbuf):
    if status == 0:
        return
    msg = errbuf.value.decode('utf-8', 'replace') if errbuf is not None else ''
    if not msg:
        msg = load().edhip_status_string(status).decode()
    raise _STATUS_EXC.get(status, RuntimeError)(msg)


_I64x8 = ctypes.c_int64 * MAX_DIMS


def describe(data_ptr, dtype_name, shape, strides_bytes):
    """Build a struct edhip_array from raw parts."""
    code = DTYPE_CODES.get(dtype_name)
    if code is None:
        # float16 / complex / ...: what the reference answers for them (deform.c:744,891)
        raise RuntimeError('data type not supported')
    nd = len(shape)
    if nd < 1 or nd > MAX_DIMS:
        raise RuntimeError('arrays must have 1..%d dimensions' % MAX_DIMS)
    return EdhipArray(data_ptr, code, nd, _I64x8(*shape), _I64x8(*strides_bytes))


class DeformArgs(object):
    """The host-side parameter arrays of one edhip_deform call (axis, orders, modes, cvals, crop
    offsets, inverse affine) converted to ctypes once; a cached Plan keeps them, so repeated calls
    with the same arguments skip the NumPy / ctypes conversions."""
    __slots__ = ("n", "naxis", "axis", "orders", "modes", "cvals", "off", "aff")

    def __init__(self, n, axis, orders, modes, cvals, output_offset, inverse_affine):
        axis = numpy.ascontiguousarray(numpy.asarray(axis, dtype=numpy.int32).reshape(n, -1))
        self.n = n
        self.naxis = int(axis.shape[1])
        self.axis = (ctypes.c_int32 * axis.size)(*[int(v) for v in axis.reshape(-1)])
        self.orders = (ctypes.c_int32 * n)(*[int(v) for v in orders])
        self.modes = (ctypes.c_int32 * n)(*[int(v) for v in modes])
        self.cvals = (ctypes.c_double * n)(*[float(v) for v in cvals])
        self.off = None
        if output_offset is not None:
            self.off = (ctypes.c_int64 * len(output_offset))(*[int(v) for v in output_offset])
        self.aff = None
        if inverse_affine is not None:
            flat = numpy.ascontiguousarray(inverse_affine, dtype=numpy.float64).reshape(-1)
            self.aff = (ctypes.c_double * flat.size)(*[float(v) for v in flat])


_errbuf = threading.local()


def _buf():
    b = getattr(_errbuf, "b", None)
    if b is None:
        b = _errbuf.b = ctypes.create_string_buffer(256)
    return b


def deform(gradient, in_descs, disp_desc, output_offset, out_descs, axis, orders, modes, cvals,
           inverse_affine, flags, stream, prepared=None, may_decline=False):
    """edhip_deform -- argument for argument `_deform_grid.deform_grid(_grad)` of the reference
    (_deform_grid.c:108-118) with descriptors in place of arrays, plus flags and the HIP stream.
    `prepared`: a DeformArgs built earlier from the same parameter arrays.  `may_decline`: return
    EDHIP_ERR_UNSUPPORTED (nothing launched) instead of raising; otherwise returns 0."""
    L = load()
    n = len(in_descs)
    a = prepared if prepared is not None else DeformArgs(n, axis, orders, modes, cvals, output_offset,
                                                         inverse_affine)
    ins = (EdhipArray * n)(*in_descs)
    outs = (EdhipArray * n)(*out_descs)
    buf = _buf()
    status = L.edhip_deform(1 if gradient else 0, n, ins, ctypes.byref(disp_desc), a.off, outs, a.naxis,
                            a.axis, a.orders, a.modes, a.cvals, a.aff, int(flags), stream, buf, 256)
    if status and not (may_decline and status == ERR_UNSUPPORTED):
        raise_for_status(status, buf)
    return status


def deform_batch(gradient, in_descs, disp_descs, output_offset, out_descs, axis, order, mode, cval,
                 inverse_affine, flags, stream):
    """edhip_deform_batch: one volume and one control grid per item, shared parameters."""
    L = load()
    n = len(in_descs)
    axis = numpy.ascontiguousarray(axis, dtype=numpy.int32).reshape(-1)
    ins = (EdhipArray * n)(*in_descs)
    disps = (EdhipArray * n)(*disp_descs)
    outs = (EdhipArray * n)(*out_descs)
    off = aff = None
    if output_offset is not None:
        off_arr = numpy.ascontiguousarray(output_offset, dtype=numpy.int64)
        off = off_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    if inverse_affine is not None:
        aff_arr = numpy.ascontiguousarray(inverse_affine, dtype=numpy.float64)
        aff = aff_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    buf = ctypes.create_string_buffer(256)
    status = L.edhip_deform_batch(
        int(bool(gradient)), n, ins, disps, off, outs, len(axis),
        axis.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(order), int(mode), float(cval), aff,
        int(flags), ctypes.c_void_p(stream), buf, 256)
    raise_for_status(status, buf)


def deform_batch_strided(gradient, nbatch, in_desc, in_bstride, disp_desc, disp_bstride, output_offset,
                         out_desc, out_bstride, axis, order, mode, cval, inverse_affine, flags, stream):
    """edhip_deform_batch_strided: the batch described once -- sample 0's descriptors plus the byte
    distance between consecutive samples of each stacked array."""
    L = load()
    axis = numpy.ascontiguousarray(axis, dtype=numpy.int32).reshape(-1)
    off = aff = None
    if output_offset is not None:
        off_arr = numpy.ascontiguousarray(output_offset, dtype=numpy.int64)
        off = off_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    if inverse_affine is not None:
        aff_arr = numpy.ascontiguousarray(inverse_affine, dtype=numpy.float64)
        aff = aff_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    buf = ctypes.create_string_buffer(256)
    status = L.edhip_deform_batch_strided(
        int(bool(gradient)), int(nbatch), ctypes.byref(in_desc), int(in_bstride), ctypes.byref(disp_desc),
        int(disp_bstride), off, ctypes.byref(out_desc), int(out_bstride), len(axis),
        axis.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(order), int(mode), float(cval), aff,
        int(flags), ctypes.c_void_p(stream), buf, 256)
    raise_for_status(status, buf)


def source_box(disp_desc, in_len, out_len, output_offset, inverse_affine, flags, stream):
    """edhip_source_box -> int64 array (naxis, 2): floor(min) / ceil(max) of the unmapped source
    coordinate along every deformed axis.  Synchronises the stream."""
    L = load()
    in_len = numpy.ascontiguousarray(in_len, dtype=numpy.int64)
    out_len = numpy.ascontiguousarray(out_len, dtype=numpy.int64)
    naxis = len(in_len)
    p64 = ctypes.POINTER(ctypes.c_int64)
    off = aff = None
    if output_offset is not None:
        off_arr = numpy.ascontiguousarray(output_offset, dtype=numpy.int64)
        off = off_arr.ctypes.data_as(p64)
    if inverse_affine is not None:
        aff_arr = numpy.ascontiguousarray(inverse_affine, dtype=numpy.float64)
        aff = aff_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    box = numpy.zeros((naxis, 2), dtype=numpy.int64)
    buf = ctypes.create_string_buffer(256)
    status = L.edhip_source_box(ctypes.byref(disp_desc), in_len.ctypes.data_as(p64),
                                out_len.ctypes.data_as(p64), off, naxis, aff, int(flags),
                                ctypes.c_void_p(stream), box.ctypes.data_as(p64), buf, 256)
    raise_for_status(status, buf)
    return box


def source_window(disp_desc, in_len, out_len, output_offset, inverse_affine, shape, axis, order, mode,
                  margin, align, minlen, flags, stream, window_ptr):
    """edhip_source_window: the filter window of one input, written to DEVICE memory at `window_ptr`
    (2 * len(shape) int32).  No synchronisation.  Returns the status (0, or EDHIP_ERR_UNSUPPORTED)."""
    L = load()
    in_len = numpy.ascontiguousarray(in_len, dtype=numpy.int64)
    out_len = numpy.ascontiguousarray(out_len, dtype=numpy.int64)
    shape = numpy.ascontiguousarray(shape, dtype=numpy.int64)
    naxis = len(in_len)
    p64 = ctypes.POINTER(ctypes.c_int64)
    off = aff = None
    if output_offset is not None:
        off_arr = numpy.ascontiguousarray(output_offset, dtype=numpy.int64)
        off = off_arr.ctypes.data_as(p64)
    if inverse_affine is not None:
        aff_arr = numpy.ascontiguousarray(inverse_affine, dtype=numpy.float64)
        aff = aff_arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    buf = _buf()
    status = L.edhip_source_window(ctypes.byref(disp_desc), in_len.ctypes.data_as(p64),
                                   out_len.ctypes.data_as(p64), off, naxis, aff, len(shape),
                                   shape.ctypes.data_as(p64), (ctypes.c_int32 * naxis)(*[int(a) for a in axis]),
                                   int(order), int(mode), int(margin), int(align), int(minlen), int(flags),
                                   ctypes.c_void_p(stream), ctypes.c_void_p(window_ptr), buf, 256)
    if status and status != ERR_UNSUPPORTED:
        raise_for_status(status, buf)
    return status


def spline_filter_axes_window(in_desc, out_desc, axes, order, transpose, window_ptr, flags, stream):
    """edhip_spline_filter_axes_window; returns the status (0, or EDHIP_ERR_UNSUPPORTED with nothing launched)"""
    L = load()
    n = len(axes)
    buf = _buf()
    status = L.edhip_spline_filter_axes_window(ctypes.byref(in_desc), ctypes.byref(out_desc), n,
                                               (ctypes.c_int32 * n)(*axes), int(order), int(bool(transpose)),
                                               ctypes.c_void_p(window_ptr), int(flags), stream, buf, 256)
    if status and status != ERR_UNSUPPORTED:
        raise_for_status(status, buf)
    return status


def spline_filter1d(in_desc, out_desc, axis, order, transpose, flags, stream):
    """edhip_spline_filter1d"""
    L = load()
    buf = _buf()
    status = L.edhip_spline_filter1d(ctypes.byref(in_desc), ctypes.byref(out_desc), int(axis),
                                     int(order), int(bool(transpose)), int(flags), stream, buf, 256)
    if status:
        raise_for_status(status, buf)


def spline_filter_axes(in_desc, out_desc, axes, order, transpose, flags, stream, may_decline=False):
    """edhip_spline_filter_axes: the whole chain (first pass in -> out, the rest in place) in one call.
    `may_decline`: return EDHIP_ERR_UNSUPPORTED (nothing launched) instead of raising; otherwise returns 0."""
    L = load()
    n = len(axes)
    buf = _buf()
    status = L.edhip_spline_filter_axes(ctypes.byref(in_desc), ctypes.byref(out_desc), n,
                                        (ctypes.c_int32 * n)(*axes), int(order), int(bool(transpose)),
                                        int(flags), stream, buf, 256)
    if status and not (may_decline and status == ERR_UNSUPPORTED):
        raise_for_status(status, buf)
    return status


def release_scratch():
    """edhip_release_scratch: free the library's cached per-stream workspaces."""
    load().edhip_release_scratch()
