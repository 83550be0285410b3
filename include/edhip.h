/*
 * edhip.h -- C ABI of the MI355X-native elastic grid deformation hot path.
 *
 * This is the drop-in boundary.  It replaces, one for one, the three entry points of the
 * reference's CPython extension `_deform_grid` (method table
 * /root/reference/elasticdeform/_deform_grid.c:306-311):
 *
 *   _deform_grid.deform_grid(...)          _deform_grid.c:296-299  -> edhip_deform(gradient=0, ...)
 *   _deform_grid.deform_grid_grad(...)     _deform_grid.c:301-304  -> edhip_deform(gradient=1, ...)
 *   _deform_grid.spline_filter1d_grad(...) _deform_grid.c:61-92    -> edhip_spline_filter1d(transpose=1)
 *   scipy.ndimage.spline_filter1d(...)     call sites deform_grid.py:160,168,271
 *                                                                  -> edhip_spline_filter1d(transpose=0)
 *
 * and its argument list mirrors the C core they all funnel into,
 *
 *   int DeformGrid(int gradient, int ninputs, PyArrayObject** inputs, PyArrayObject* displacement,
 *                  PyArrayObject* output_offset, PyArrayObject** outputs, int naxis, int* axis,
 *                  int* orders, int* modes, double* cvals, double* affine);      deform.h:15-16
 *   int NI_SplineFilter1DGrad(PyArrayObject*, int order, int axis, PyArrayObject*);  deform.h:18
 *
 * with `PyArrayObject*` replaced by the plain-old-data descriptor `edhip_array`
 * (pointer + dtype code + shape + byte strides == what PyArray_DATA / PyArray_TYPE /
 * PyArray_DIM / PyArray_STRIDE give the reference, deform.c:383-391,401,427-431).
 *
 * No Python, NumPy or torch type appears in any signature.  `data` pointers are DEVICE pointers
 * (HBM on the MI355X that owns `hip_stream`); the small parameter arrays (axis, orders, modes,
 * cvals, affine, output_offset) are HOST pointers, read before the call returns.
 *
 * Ownership: the caller allocates every array; the library borrows them for the duration of the
 * enqueued work and owns only a small scratch workspace per (device, stream), grown on demand.
 * Calls are asynchronous with respect to the host: work is enqueued on `hip_stream` and the
 * function returns; there is no implicit device synchronisation -- with one exception: when a call
 * needs MORE scratch than the stream's cached workspace holds, growing it waits for that stream once
 * (hipStreamSynchronize + hipFree + hipMalloc; not allowed under stream capture: warm the workspace
 * up with one call before capturing).  Requests above 320 MiB (the dense temporary of the float32
 * order-4/5 prefilter cascade on long lines, the fp64 line buffer of the exact filter on large
 * arrays) are stream-ordered allocations that are released when the call has enqueued its work.
 * Next to the workspace the library keeps, per (device, stream), 64 bytes of pinned host memory into
 * which the float32 tile kernels report how many of a call's tiles did not fit their standard LDS boxes
 * (written by the NEXT call's first kernel, read by the host without waiting): later calls of the same
 * geometry then start on larger boxes.  This changes speed only -- the forward result of a voxel is the
 * same bits whichever box size or tile level served it.  Apart from these mutex-guarded per-stream
 * caches the library keeps no mutable global state, so it is re-entrant and may be
 * called concurrently from several host threads on different streams / devices
 * (reference: single-threaded, GIL released for the whole call, deform.c:377-379).
 *
 * Error convention (reference: return 0 with a Python exception set, deform.c:1042,
 * _deform_grid.c:293): return EDHIP_OK (0) on success, otherwise a non-zero code and a
 * NUL-terminated message in `err` (if errlen > 0).  The host shim maps codes to the exception
 * classes the reference raises.
 */
#ifndef EDHIP_H
#define EDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDHIP_VERSION 100        /* 0.1.0 */
#define EDHIP_MAX_DIMS 8         /* max ndim of any array (reference: NPY_MAXDIMS) */
#define EDHIP_MAX_AXES 7         /* max number of deformed axes: the control grid has naxis + 1 <= EDHIP_MAX_DIMS dims */
#define EDHIP_MAX_INPUTS 64

/* dtype codes: the 13 NumPy types the reference switches over (deform.c:716-741,863-887,
 * 907-919) collapse to 11 distinct machine types on LP64 (long == long long); two 16-bit floating
 * point storage types are an extension. */
enum edhip_dtype {
    EDHIP_BOOL = 0, /* npy_bool  (unsigned char; store is a plain C cast, deform.c:287-290,907) */
    EDHIP_U8 = 1,
    EDHIP_I8 = 2,
    EDHIP_U16 = 3,
    EDHIP_I16 = 4,
    EDHIP_U32 = 5,
    EDHIP_I32 = 6,
    EDHIP_U64 = 7,
    EDHIP_I64 = 8,
    EDHIP_F32 = 9,
    EDHIP_F64 = 10,
    /* reduced-precision storage -- NOT part of the reference's
     * contract, which rejects half precision with "data type not supported" (deform.c:742-747); the
     * host shim only hands these over when the caller opted in.  Arithmetic stays fp64 / fp32:
     * values are widened on load and rounded to nearest-even on store. */
    EDHIP_F16 = 11, /* IEEE binary16 */
    EDHIP_BF16 = 12, /* bfloat16 */
    EDHIP_NUM_DTYPES = 13
};

/* boundary modes, same integer codes as the reference (from_scipy.h:38-47,
 * deform_grid.py:440-454) */
enum edhip_mode {
    EDHIP_MODE_NEAREST = 0,
    EDHIP_MODE_WRAP = 1,
    EDHIP_MODE_REFLECT = 2,
    EDHIP_MODE_MIRROR = 3,
    EDHIP_MODE_CONSTANT = 4
};

/* return codes */
enum edhip_status {
    EDHIP_OK = 0,
    EDHIP_ERR_INVALID = 1,     /* shape / argument check failed   -> RuntimeError  (_deform_grid.c:121-255) */
    EDHIP_ERR_DTYPE = 2,       /* "data type not supported"       -> RuntimeError  (deform.c:744,891,922)   */
    EDHIP_ERR_MEMORY = 3,      /* scratch allocation failed       -> MemoryError   (deform.c:394-398 ...)   */
    EDHIP_ERR_DEVICE = 4,      /* a HIP call / kernel launch failed -> RuntimeError                         */
    EDHIP_ERR_UNSUPPORTED = 5  /* legal in the reference, outside this build's limits (more than 8 dims ...)   */
};

/* arithmetic selection for edhip_deform / edhip_spline_filter1d */
enum edhip_flags {
    EDHIP_FLAG_AUTO = 0,       /* float32 / float64 data -> fast path; integer and bool data -> bit-equal to the
                                  exact path (order-0 label maps and 8- / 16- / 32-bit integer volumes of orders 1-5: fast
                                  coordinates, near-tie voxels re-evaluated exactly; the rest: exact kernels) */
    EDHIP_FLAG_EXACT = 1,      /* fp64 arithmetic in the reference's own evaluation order (bit-comparable) */
    EDHIP_FLAG_FAST = 2,       /* fp64 coordinates, restructured (separable) sums, data-width tap accumulation
                                  (what AUTO selects for floating-point data).
                                  Given explicitly it also opens the pairs of a float32 array and a 16-bit float one
                                  (EDHIP_F16 / EDHIP_BF16; an extension -- the reference rejects float16, deform.c:742-747):
                                  float32 arithmetic with 16 bits in HBM on one side, converted inside the kernel that
                                  reads or writes it instead of by a cast pass --
                                    edhip_deform  inputs float32, outputs 16-bit: forward = the result is rounded to nearest
                                                  even at the store; gradient = dY is widened at the load (dX stays float32);
                                                  3 deformed axes, orders 1-3, unit stride along the last one;
                                    edhip_spline_filter_axes  16-bit input, float32 output: the first pass widens;
                                                  float32 input, 16-bit output with EDHIP_FLAG_SCRATCH_INPUT: the last pass
                                                  narrows; orders 2 / 3, lines of 64..256 samples, dense arrays.
                                  A pair outside those envelopes returns EDHIP_ERR_UNSUPPORTED
                                  (without the flag such pairs run on the exact kernels like any other dtype pair).
                                  edhip_deform decides per input: the control-grid prefilter of a RAW_DISPLACEMENT call
                                  (and the clear of a ZERO_GRADIENT call) has been enqueued by then, and with several
                                  inputs the kernels of the inputs in front of the declined one have run -- a caller
                                  that retries another way must treat every output of the call as unwritten and zero
                                  gradient accumulators again.  The filter entry points decline before any launch. */
    /* edhip_deform only: `displacement` is the RAW control grid; the library applies the order-3
     * mirror prefilter along every grid axis itself (what deform_grid.py:166-169,269-272 does with
     * SciPy before calling the C code), in one launch, with the same arithmetic and the same
     * per-axis rounding to the grid's dtype.  Grids of more than 4096 points are refused
     * (EDHIP_ERR_UNSUPPORTED): prefilter those with edhip_spline_filter1d. */
    EDHIP_FLAG_RAW_DISPLACEMENT = 4,
    /* edhip_deform / edhip_deform_batch*, forward (gradient = 0): also leave the bounding boxes of
     * the tiles' tap windows -- a by-product of the forward kernel -- in a small per-stream buffer
     * for the gradient call that follows (float32, 3 deformed axes: the tile kernels of
     * deform_hot.hip; ignored elsewhere). */
    EDHIP_FLAG_KEEP_BOXES = 8,
    /* gradient = 1: the caller promises that the displacement's CONTENTS and the geometry are those of
     * the last EDHIP_FLAG_KEEP_BOXES forward call on this stream (autograd: the backward of that
     * forward).  The library compares the arguments it can see (pointer, shape, strides, extents,
     * offsets, affine, order, mode); when they match the gradient kernel takes its tiles' boxes from
     * the buffer instead of computing every window twice.  The boxes are a HINT: a voxel whose window
     * does not lie in its tile's box is scattered straight to global memory, so a broken promise
     * costs time, never correctness. */
    EDHIP_FLAG_USE_BOXES = 16,
    /* gradient = 1: the library clears the gradient accumulators (`inputs`) itself before it scatters
     * into them -- the reference's numpy.zeros (deform_grid.py:243) -- so the caller may hand over
     * uninitialised memory.  The arrays must be dense (their elements one contiguous block, in any axis
     * order); float32 volumes on the tile kernels get the fill from spare workgroups of the per-call tables
     * launch (it overlaps that kernel instead of being a bandwidth-bound launch of its own), every other
     * route a hipMemsetAsync.  Without the flag the accumulators are added to as they are. */
    EDHIP_FLAG_ZERO_GRADIENT = 32,
    /* with EDHIP_FLAG_RAW_DISPLACEMENT: the caller promises that the raw control grid is, byte for byte, the one
     * of the previous RAW_DISPLACEMENT call on this stream (edhip_source_window followed by edhip_deform inside
     * one deform_grid call): when the library finds that call's filtered copy still in the stream's workspace
     * (same pointer, dtype, shape and strides, workspace not moved) it does not filter the grid again. */
    EDHIP_FLAG_GRID_STAYS = 64,
    /* edhip_spline_filter_axes: the caller hands `input` over as scratch -- every pass but the last runs in place on
     * it and the last one writes `output` (instead of: input -> output, then in place on the output).  With a
     * float32 input and an EDHIP_F16 / EDHIP_BF16 output the chain is computed in float32 and only its result is
     * rounded to 16 bits (the last pass narrows on its way out: 6 instead of 8 bytes per sample, and no cast pass). */
    EDHIP_FLAG_SCRATCH_INPUT = 128,
    /* edhip_deform / edhip_deform_batch*, forward, float32 volumes, 3 deformed axes, orders 1-3: a HINT that the
     * displacement field is strong (displacement gradient of ~0.15 per voxel and more: sigma >= 10 on a 5^3 grid over
     * 256^3, the reference README's own example).  The forward call then takes the z-walk route (csrc/deform_k1z.hip:
     * per-call geometry kernel, tiles split in halves, four row pitches) for EVERY geometry that route supports; without
     * the flag only single volumes whose z and y extents are multiples of 256 take it -- where it is measured faster
     * on mild fields too (profiles/r06_k1_route_sweep.txt) -- and everything else runs on the x-strip kernel
     * (csrc/deform_k1.hip), which is 5-50 % faster on mild fields and 5-35 % slower on strong ones.  The route is a
     * function of the call's arguments alone, never of earlier calls; the two routes agree to float32 rounding
     * (coordinates differ by ~1e-15), each is bit-reproducible. */
    EDHIP_FLAG_STRONG_FIELD = 256
};

/* strided N-d array in device memory: the POD stand-in for PyArrayObject* */
typedef struct edhip_array {
    void*   data;                          /* device pointer to element [0,...,0]          */
    int32_t dtype;                         /* enum edhip_dtype                              */
    int32_t ndim;                          /* 0 < ndim <= EDHIP_MAX_DIMS                    */
    int64_t shape[EDHIP_MAX_DIMS];         /* extents                                       */
    int64_t stride_bytes[EDHIP_MAX_DIMS];  /* byte strides (any sign, any order, may be 0)  */
} edhip_array;

/* library version (EDHIP_VERSION of the build) */
int edhip_version(void);

/* static string for a status code */
const char* edhip_status_string(int status);

/* number of visible HIP devices, or -1 if the HIP runtime cannot be initialised.  Lets a host
 * shim fail loudly before any compute call. */
int edhip_device_count(void);

/*
 * Forward deformation (gradient == 0) or its exact adjoint (gradient != 0).
 * Replaces DeformGrid (deform.c:340-1043) behind Py_DeformGrid_helper (_deform_grid.c:94-294).
 *
 *   inputs[ninputs]      forward: (prefiltered) source arrays, read.
 *                        gradient: dX arrays, accumulated into -- MUST be zero-filled on entry
 *                        (deform_grid.py:243).
 *   displacement         prefiltered control-point grid, shape (naxis, ncp_0, ..., ncp_{naxis-1}),
 *                        any dtype / strides (deform.c:388-391,715-741).
 *   output_offset        naxis crop offsets (host int64) or NULL (deform.c:439-446).
 *   outputs[ninputs]     forward: written.  gradient: dY arrays, read.
 *                        outputs[i].ndim == inputs[i].ndim; deformed extents of every output equal
 *                        those of outputs[0], of every input those of inputs[0]
 *                        (_deform_grid.c:137-175).
 *   axis                 host int32[ninputs * naxis], the deformed axes of each input, ascending.
 *   orders, modes, cvals host arrays of length ninputs: spline order 0..5, enum edhip_mode, cval.
 *   affine               host double[naxis * (naxis+1)], row-major INVERSE map output->source,
 *                        or NULL (deform.c:771-776; built by deform_grid.py:392-438).
 *   flags                enum edhip_flags.
 *   hip_stream           hipStream_t to enqueue on (NULL = the legacy default stream).
 *
 * Float gradients (fast arithmetic, 3 deformed axes): a tile of output voxels is accumulated in LDS
 * in fixed point, in a scale taken from the tile's sum of |dY|; one contribution is resolved to
 * wmax * sum|dY| / 2^31 of its tile (float32; 2^62 for float64) -- far below the float32 rounding of
 * the reference's own `+=` (deform.c:309-312) for data of one magnitude.  The bound is ABSOLUTE per
 * tile (8 x 8 x 16 output voxels): next to a dY value that is orders of magnitude above its
 * neighbours, the neighbours' contributions are rounded to that resolution (1e6 among 1e-3: the small
 * ones keep about 1e-4 absolute), where the reference's float `+=` keeps their relative precision in
 * cells the large value does not reach; other tiles are unaffected.  Non-finite dY values are added
 * with float atomics.  The order of the atomic additions is not reproducible from run to run
 * (about 1e-7 of the gradient's scale).  Integer gradients are bit-exact.
 */
int edhip_deform(int gradient, int ninputs,
                 const edhip_array* inputs,
                 const edhip_array* displacement,
                 const int64_t* output_offset,
                 const edhip_array* outputs,
                 int naxis, const int32_t* axis,
                 const int32_t* orders, const int32_t* modes, const double* cvals,
                 const double* affine,
                 uint32_t flags, void* hip_stream,
                 char* err, size_t errlen);

/*
 * A batch of independent volumes, each with its own control grid (per-sample augmentation:
 * SURVEY.md section 8(f) rank 2): item b is exactly
 *   edhip_deform(gradient, 1, &inputs[b], &displacements[b], output_offset, &outputs[b], naxis,
 *                axis, &order, &mode, &cval, affine, flags, hip_stream, ...)
 * -- same results, bit for bit.  `axis` (naxis entries), order, mode, cval, the crop offsets and
 * the affine map are shared by the batch.  Descriptors that differ only by a constant pointer
 * stride (see edhip_deform_batch_strided) take the single-launch path described there; anything
 * else is enqueued item by item.
 * The reference has no batched entry point; a host loop over its deform_grid is the equivalent.
 */
int edhip_deform_batch(int gradient, int nbatch,
                       const edhip_array* inputs,
                       const edhip_array* displacements,
                       const int64_t* output_offset,
                       const edhip_array* outputs,
                       int naxis, const int32_t* axis,
                       int32_t order, int32_t mode, double cval,
                       const double* affine,
                       uint32_t flags, void* hip_stream,
                       char* err, size_t errlen);

/*
 * The same batch described once: sample b's arrays are sample 0's moved by b * stride bytes (a
 * stacked tensor with a leading batch axis).  Spares the caller nbatch descriptors per array.
 * When the volumes are float32 / float64 with 3 deformed axes and the control grids are already
 * prefiltered (no EDHIP_FLAG_RAW_DISPLACEMENT), the whole batch is ONE set of launches: one tables
 * kernel producing the per-sample displacement tables, one tile launch whose strip index carries
 * the sample, one pair of spill passes -- bit-identical to nbatch edhip_deform calls.
 */
int edhip_deform_batch_strided(int gradient, int nbatch,
                               const edhip_array* input0, int64_t input_batch_stride,
                               const edhip_array* displacement0, int64_t displacement_batch_stride,
                               const int64_t* output_offset,
                               const edhip_array* output0, int64_t output_batch_stride,
                               int naxis, const int32_t* axis,
                               int32_t order, int32_t mode, double cval,
                               const double* affine,
                               uint32_t flags, void* hip_stream,
                               char* err, size_t errlen);

/*
 * Frees the scratch workspaces the library caches per (device, stream): per-call tables, spill
 * lists, the fp64 line buffers of the exact prefilter and the dense temporary of the order-4/5
 * cascade (up to the size of the largest array filtered that way).  Waits for the owning devices to
 * go idle first.  Later calls allocate again on demand.  Must not run concurrently with other
 * calls into the library.
 */
int edhip_release_scratch(void);

/*
 * Measurement aid (bench.py): with profiling enabled, edhip_deform brackets the launch of its
 * dominant kernel -- the LDS-tiled forward / gradient kernel over all strips, without the tables
 * kernel and the spill passes -- with HIP events recorded on `hip_stream`.
 * edhip_profile_last_us() waits for the most recent such launch of the calling thread and returns
 * its duration in microseconds (-1 if there is none).  Off by default; no effect on results.
 */
int edhip_profile_dominant(int enable);
double edhip_profile_last_us(void);

/*
 * Bounding box of the source coordinates of a call: for every deformed axis h,
 *   box[2h]   = floor(min c_h),   box[2h+1] = ceil(max c_h)
 * over all output voxels, where c_h = affine(o)_h + offset_h + displacement_h(o) is the source
 * coordinate of deform.c:771-781 BEFORE the boundary map (values are clamped to +-1e9).  The
 * arguments have the meaning they have in edhip_deform; in_len / out_len (host int64[naxis]) are
 * the deformed extents of inputs[0] / outputs[0].  The box is returned to the host, so this call
 * SYNCHRONISES `hip_stream`.  Control grids of more than 7680 values (naxis * prod ncp) are
 * refused with EDHIP_ERR_UNSUPPORTED.  With EDHIP_FLAG_FAST the box is the conservative one of the
 * spline's convex hull -- [min, max] of the control coefficients that reach the output box, refined by up
 * to two levels of B-spline subdivision, plus the affine part at the box's corners: a superset of the exact
 * box (about 1.2x the displacement's true range for a random 5^3 grid; the unrefined hull is 9x), computed
 * from the control grid alone in microseconds, where the exact scan visits every output voxel.
 *
 * No counterpart in the reference.  It lets the host layer restrict the input prefilter
 * (deform_grid.py:155-164) to the part of a volume that a cropped output can reach.
 */
int edhip_source_box(const edhip_array* displacement,
                     const int64_t* in_len, const int64_t* out_len,
                     const int64_t* output_offset, int naxis, const double* affine,
                     uint32_t flags, void* hip_stream,
                     int64_t* box,
                     char* err, size_t errlen);

/*
 * One-dimensional B-spline prefilter along `axis`, mirror boundary.
 *   transpose == 0 : scipy.ndimage.spline_filter1d(input, order, axis, output, mode='mirror')
 *                    (call sites deform_grid.py:160,168,271)
 *   transpose != 0 : its exact adjoint, NI_SplineFilter1DGrad (deform.c:1049-1168)
 * input/output: same shape, any of the 11 dtypes each, may alias (in-place) -- the reference calls
 * it in place from the second axis on (deform_grid.py:158-161,280-283).  order 0/1: plain copy
 * with dtype conversion.  Arithmetic is fp64; the result is rounded to the output dtype once per
 * call, exactly like the reference's double line buffer (deform.c:1106-1162).
 */
int edhip_spline_filter1d(const edhip_array* input, const edhip_array* output,
                          int axis, int order, int transpose,
                          uint32_t flags, void* hip_stream,
                          char* err, size_t errlen);

/*
 * The filter chain of one array in one call: edhip_spline_filter1d along axes[0] from `input` into
 * `output`, then along axes[1..] in place in `output` -- the loop of deform_grid.py:157-162 (forward)
 * and :279-284 (transpose).  input may equal output (every pass in place).  One host call instead of
 * naxes (the per-call host overhead matters for small volumes).
 */
int edhip_spline_filter_axes(const edhip_array* input, const edhip_array* output,
                             int naxes, const int32_t* axes, int order, int transpose,
                             uint32_t flags, void* hip_stream,
                             char* err, size_t errlen);

/*
 * Crop-aware prefilter with the window kept ON THE DEVICE (no read-back, no stream synchronisation).
 * The reference prefilters every input whole before it deforms it (deform_grid.py:155-164) and runs the
 * transposed filter over the whole gradient afterwards (:277-286); with a crop only a box of the input is
 * ever read.  edhip_source_window computes, for ONE input of an edhip_deform call with these displacement /
 * crop / affine arguments, the index window [w0, w1) along each of the input's deformed axes outside which no
 * tap of the output can fall -- the range of the source coordinate (a superset: convex hull of the subdivided
 * control coefficients), the tap window of deform.c:783-813, the boundary mode's clipping or folding
 * (deform.c:47-128: 'nearest' / 'constant' clip, the folding modes keep the whole axis), `margin` samples of
 * filter decay on either side -- and writes 2 * ndim int32 (w0, w1) in the input's dimension order into
 * `window` (DEVICE memory; non-deformed dimensions get (0, shape[d])).  Rows of the last dimension start and
 * end on multiples of `align` elements; a deformed axis keeps at least `minlen` samples.
 * edhip_spline_filter_axes_window is edhip_spline_filter_axes restricted to that window: samples outside it
 * are neither read nor written, lines are filtered as lines of the window's length.  Whole-line tile kernels
 * only (float32 / float64, orders 2 / 3, lines of 64 .. 576 samples): EDHIP_ERR_UNSUPPORTED, with nothing
 * launched, when a pass is outside that envelope -- filter the whole array then.
 */
int edhip_source_window(const edhip_array* displacement, const int64_t* in_len, const int64_t* out_len,
                        const int64_t* output_offset /* naxis or NULL */, int naxis,
                        const double* inv_affine /* naxis*(naxis+1) or NULL */,
                        int ndim, const int64_t* shape /* ndim: the input's extents */,
                        const int32_t* axis /* naxis: the input's deformed dimensions */, int order, int mode,
                        int margin, int align, int minlen, uint32_t flags, void* hip_stream,
                        int32_t* window /* device, 2 * ndim */, char* err, size_t errlen);
int edhip_spline_filter_axes_window(const edhip_array* input, const edhip_array* output, int naxes,
                                    const int32_t* axes, int order, int transpose,
                                    const int32_t* window /* device, 2 * ndim */, uint32_t flags,
                                    void* hip_stream, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif /* EDHIP_H */
