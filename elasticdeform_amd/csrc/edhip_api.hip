// edhip_api.hip -- the C ABI of include/edhip.h: argument validation (the checks of
// Py_DeformGrid_helper, _deform_grid.c:121-255, and Py_SplineFilter1D_grad, :71-81), kernel
// selection and stream-ordered launches.  No Python, NumPy or torch types.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ed_device.h"
#include "ed_params.h"
#include "ed_workspace.h"
#include "edhip.h"

namespace {

int fail(char* err, size_t errlen, int code, const char* fmt, ...)
{
    if (err && errlen) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, errlen, fmt, ap);
        va_end(ap);
    }
    return code;
}

int hip_fail(char* err, size_t errlen, hipError_t e, const char* what)
{
    if (e == hipErrorOutOfMemory)
        return fail(err, errlen, EDHIP_ERR_MEMORY, "%s: out of device memory", what);
    return fail(err, errlen, EDHIP_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

bool dtype_ok(int dt) { return dt >= 0 && dt < EDHIP_NUM_DTYPES; }

int dtype_size(int dt)
{
    switch (dt) {
    case EDHIP_BOOL: case EDHIP_U8: case EDHIP_I8: return 1;
    case EDHIP_U16: case EDHIP_I16: case EDHIP_F16: case EDHIP_BF16: return 2;
    case EDHIP_U32: case EDHIP_I32: case EDHIP_F32: return 4;
    default: return 8;
    }
}

int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

// Geometry shared by every input of a call (deform.c:381-391,439-451,771-776); with
// EDHIP_FLAG_RAW_DISPLACEMENT the control grid is prefiltered into the head of the workspace.
// zero_ptr / zero_bytes / zero_done: a gradient block the grid-prefilter launch may clear on its spare workgroups
// (*zero_done says whether it did: only a RAW_DISPLACEMENT call that really launches the prefilter)
// defer: the caller enqueues the grid prefilter itself (edhip_deform: the tables kernel of the tile path does it in its
// own launch) -- *defer gets the parameters, *defer_stamp what the stream's stamp becomes once the filter is enqueued;
// defer->total stays 0 when there is nothing to filter (no RAW_DISPLACEMENT, or the filtered copy is still there)
int make_geometry(const edhip_array* displacement, const int64_t* in_len, const int64_t* out_len,
                  const int64_t* output_offset, int naxis, const double* affine, uint32_t flags,
                  hipStream_t stream, ed::GridGeom& g, char* err, size_t errlen, char* zero_ptr = nullptr,
                  long long zero_bytes = 0, bool* zero_done = nullptr, ed::GridPrefilter* defer = nullptr,
                  ed::GridStamp* defer_stamp = nullptr, bool f64_volumes = false)
{
    using namespace ed;
    memset(&g, 0, sizeof(g));
    g.naxis = naxis;
    g.has_affine = affine != nullptr;
    g.disp_dtype = displacement->dtype;
    g.disp = (const char*)displacement->data;
    g.nvox = 1;
    for (int k = 0; k < naxis; ++k) {
        g.in_len[k] = in_len[k];
        g.out_len[k] = out_len[k];
        g.off[k] = output_offset ? output_offset[k] : 0;
        g.ncp[k] = displacement->shape[k + 1];
        g.nvox *= g.out_len[k];
        if (g.out_len[k] > 0 && g.in_len[k] < 2)
            // the reference divides by (I_k - 1) (deform.c:643,655): undefined there, refused here
            return fail(err, errlen, EDHIP_ERR_INVALID,
                        "deformed axes must have at least 2 elements");
    }
    for (int k = 0; k <= naxis; ++k)
        g.disp_stride[k] = displacement->stride_bytes[k];
    if (flags & EDHIP_FLAG_RAW_DISPLACEMENT) {
        // prefilter the raw control grid into the head of the stream's workspace (one launch)
        int64_t total = 1;
        for (int k = 0; k <= naxis; ++k)
            total *= displacement->shape[k];
        if (total > 4096)
            return fail(err, errlen, EDHIP_ERR_UNSUPPORTED,
                        "raw displacement grids are limited to 4096 points");
        // reserve everything this call can need now: a later, larger request would move the buffer
        // and lose the grid
        hipError_t e = hipSuccess;
        void* ws = workspace_reserve(stream, deform_tile_workspace_bytes(g, 1, f64_volumes), &e);
        if (!ws)
            return hip_fail(err, errlen, e, "scratch allocation");
        GridPrefilter gp;
        memset(&gp, 0, sizeof(gp));
        gp.in = (const char*)displacement->data;
        gp.out = (char*)ws;
        gp.dtype = displacement->dtype;
        gp.elem_size = dtype_size(displacement->dtype);
        gp.ndim = naxis + 1;
        gp.total = (int)total;
        gp.pole = -0.267949192431122706472553658494127633;      // order 3, SciPy's literal
        gp.gain = (1.0 - gp.pole) * (1.0 - 1.0 / gp.pole);
        int64_t stride = gp.elem_size;
        for (int k = naxis; k >= 0; --k) {
            gp.shape[k] = (int)displacement->shape[k];
            gp.stride_bytes[k] = displacement->stride_bytes[k];
            gp.pole_pow[k] = std::pow(gp.pole, (double)(gp.shape[k] - 1));
            g.disp_stride[k] = stride;
            stride *= gp.shape[k];
        }
        // (EDHIP_FLAG_GRID_STAYS: the previous call's filtered copy of this very grid is still there)
        GridStamp* stamp = grid_stamp(stream);
        GridStamp now;
        now.raw = displacement->data;
        now.ws = ws;
        now.generation = workspace_generation(stream);
        now.dtype = displacement->dtype;
        now.ndim = naxis + 1;
        for (int k = 0; k <= naxis; ++k) {
            now.shape[k] = displacement->shape[k];
            now.stride[k] = displacement->stride_bytes[k];
        }
        const bool stays = (flags & EDHIP_FLAG_GRID_STAYS) && stamp->raw == now.raw && stamp->ws == now.ws &&
                           stamp->generation == now.generation &&
                           stamp->dtype == now.dtype && stamp->ndim == now.ndim &&
                           memcmp(stamp->shape, now.shape, sizeof(now.shape)) == 0 &&
                           memcmp(stamp->stride, now.stride, sizeof(now.stride)) == 0;
        if (!stays && defer && defer_stamp) {
            *stamp = GridStamp();
            *defer = gp;
            *defer_stamp = now;
        } else if (!stays) {
            *stamp = GridStamp();
            if (zero_ptr && zero_done && ((uintptr_t)zero_ptr & 15) == 0) {
                gp.zero_ptr = zero_ptr;
                gp.zero_bytes = zero_bytes;
                *zero_done = true;
            }
            e = launch_grid_prefilter(gp, stream);
            if (e != hipSuccess)
                return hip_fail(err, errlen, e, "grid prefilter launch");
            *stamp = now;
        }
        g.disp = (const char*)ws;
    }
    if (affine)
        for (int k = 0; k < naxis * (naxis + 1); ++k)
            g.affine[k] = affine[k];
    return EDHIP_OK;
}

// One input/output pair as the kernels see it: strides of the deformed axes, the non-deformed
// ("step") axes flattened (deform.c:399-436).
int make_view(const edhip_array& in, const edhip_array& out, int naxis, const int32_t* axis, int order,
              int mode, double cval, ed::IOView& v, char* err, size_t errlen)
{
    using namespace ed;
    memset(&v, 0, sizeof(v));
    v.in = (const char*)in.data;
    v.out = (char*)out.data;
    v.in_dtype = in.dtype;
    v.out_dtype = out.dtype;
    v.order = order;
    v.mode = mode;
    v.cval = cval;
    v.nsteps = 1;
    int64_t min_deform = INT64_MAX, min_step = INT64_MAX;
    for (int d = 0; d < in.ndim; ++d) {
        int k = -1;
        for (int j = 0; j < naxis; ++j)
            if (axis[j] == d)
                k = j;
        if (k >= 0) {
            v.in_stride[k] = in.stride_bytes[d];
            v.out_stride[k] = out.stride_bytes[d];
            if (out.shape[d] > 1 && iabs64(out.stride_bytes[d]) < min_deform)
                min_deform = iabs64(out.stride_bytes[d]);
        } else {
            if (in.shape[d] != out.shape[d])
                return fail(err, errlen, EDHIP_ERR_INVALID,
                            "non-deformed axes of input and output must have the same size");
            v.step_len[v.nstep] = in.shape[d];
            v.in_step_stride[v.nstep] = in.stride_bytes[d];
            v.out_step_stride[v.nstep] = out.stride_bytes[d];
            v.nsteps *= in.shape[d];
            if (out.shape[d] > 1 && iabs64(out.stride_bytes[d]) < min_step)
                min_step = iabs64(out.stride_bytes[d]);
            v.nstep++;
        }
    }
    v.steps_fastest = v.nstep > 0 && min_step < min_deform;
    return EDHIP_OK;
}

// descriptors of a batch that differ only in their base pointers, at a constant distance
bool uniform_batch(const edhip_array* a, int n, int64_t* stride)
{
    *stride = n > 1 ? (const char*)a[1].data - (const char*)a[0].data : 0;
    for (int b = 0; b < n; ++b) {
        if (a[b].dtype != a[0].dtype || a[b].ndim != a[0].ndim)
            return false;
        for (int d = 0; d < a[0].ndim; ++d)
            if (a[b].shape[d] != a[0].shape[d] || a[b].stride_bytes[d] != a[0].stride_bytes[d])
                return false;
        if ((const char*)a[b].data != (const char*)a[0].data + (int64_t)b * *stride)
            return false;
    }
    // the single-launch kernels address samples in elements: the distance must be a whole number of them
    const int64_t es = (a[0].dtype == EDHIP_F64 || a[0].dtype == EDHIP_I64 || a[0].dtype == EDHIP_U64) ? 8
                       : (a[0].dtype == EDHIP_F32 || a[0].dtype == EDHIP_I32 || a[0].dtype == EDHIP_U32) ? 4 : 1;
    if (*stride % es != 0)
        return false;
    return true;
}

}  // namespace

extern "C" {

#ifdef EDHIP_EXPERIMENTS
// marks a profiling build (make EXPERIMENTS=1): environment switches are live in it
int edhip_experiments_build = 1;
#endif

int edhip_version(void) { return EDHIP_VERSION; }

const char* edhip_status_string(int status)
{
    switch (status) {
    case EDHIP_OK: return "ok";
    case EDHIP_ERR_INVALID: return "invalid argument";
    case EDHIP_ERR_DTYPE: return "data type not supported";
    case EDHIP_ERR_MEMORY: return "out of memory";
    case EDHIP_ERR_DEVICE: return "HIP runtime error";
    case EDHIP_ERR_UNSUPPORTED: return "not supported by this build";
    default: return "unknown status";
    }
}

// the array's elements form one contiguous block (any axis order, no gaps, no overlap): its start and size
static bool dense_block(const edhip_array& a, char** ptr, long long* bytes)
{
    const long long esz = dtype_size(a.dtype);
    if (esz <= 0)
        return false;
    int order[EDHIP_MAX_DIMS], n = 0;
    long long total = esz;
    for (int d = 0; d < a.ndim; ++d) {
        if (a.shape[d] == 0) {
            *ptr = (char*)a.data;
            *bytes = 0;
            return true;
        }
        total *= a.shape[d];
        if (a.shape[d] > 1)
            order[n++] = d;
    }
    for (int i = 1; i < n; ++i)            // by ascending stride
        for (int j = i; j > 0 && a.stride_bytes[order[j]] < a.stride_bytes[order[j - 1]]; --j) {
            const int t = order[j];
            order[j] = order[j - 1];
            order[j - 1] = t;
        }
    long long expect = esz;
    for (int i = 0; i < n; ++i) {
        if (a.stride_bytes[order[i]] != expect)
            return false;
        expect *= a.shape[order[i]];
    }
    *ptr = (char*)a.data;
    *bytes = total;
    return true;
}

int edhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return -1;
    return n;
}

int edhip_deform(int gradient, int ninputs, const edhip_array* inputs,
                 const edhip_array* displacement, const int64_t* output_offset,
                 const edhip_array* outputs, int naxis, const int32_t* axis, const int32_t* orders,
                 const int32_t* modes, const double* cvals, const double* affine, uint32_t flags,
                 void* hip_stream, char* err, size_t errlen)
{
    using namespace ed;
    hipStream_t stream = (hipStream_t)hip_stream;
    StreamGuard guard(stream);      // the stream's scratch is ours until every launch is enqueued
    if (err && errlen)
        err[0] = 0;
    // ---- the checks of _deform_grid.c:121-255 -------------------------------------------------
    if (!inputs || !outputs || ninputs <= 0 || ninputs > EDHIP_MAX_INPUTS)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid number of inputs/outputs");
    if (!axis || !orders || !modes || !cvals || naxis < 1)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis list");
    if (naxis > kMaxAxes)
        return fail(err, errlen, EDHIP_ERR_UNSUPPORTED,
                    "more than %d deformed axes are not supported on the GPU", kMaxAxes);
    for (int i = 0; i < ninputs; ++i) {
        const edhip_array& in = inputs[i];
        const edhip_array& out = outputs[i];
        if (in.ndim != out.ndim)
            return fail(err, errlen, EDHIP_ERR_INVALID, "input and output dimensions should match");
        if (in.ndim < 1 || in.ndim > EDHIP_MAX_DIMS)
            return fail(err, errlen, EDHIP_ERR_UNSUPPORTED, "arrays must have 1..%d dimensions",
                        EDHIP_MAX_DIMS);
        if (!dtype_ok(in.dtype) || !dtype_ok(out.dtype))
            return fail(err, errlen, EDHIP_ERR_DTYPE, "data type not supported");
        for (int j = 0; j < naxis; ++j) {
            const int a = axis[i * naxis + j];
            if (a < 0 || a >= in.ndim)
                return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis in axis list");
            if (in.shape[a] != inputs[0].shape[axis[j]])
                return fail(err, errlen, EDHIP_ERR_INVALID, "all inputs should have the same size");
            if (out.shape[a] != outputs[0].shape[axis[j]])
                return fail(err, errlen, EDHIP_ERR_INVALID, "all outputs should have the same size");
        }
        if (orders[i] < 0 || orders[i] > 5)
            return fail(err, errlen, EDHIP_ERR_INVALID, "spline order not supported");
        if (modes[i] < 0 || modes[i] > 4)
            return fail(err, errlen, EDHIP_ERR_INVALID, "boundary mode not supported");
    }
    if (!displacement || displacement->ndim != naxis + 1 || displacement->shape[0] != naxis)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid displacement shape");
    if (!dtype_ok(displacement->dtype))
        return fail(err, errlen, EDHIP_ERR_DTYPE, "data type not supported");
    for (int k = 0; k <= naxis; ++k)
        if (displacement->shape[k] <= 0)
            return fail(err, errlen, EDHIP_ERR_INVALID, "invalid displacement shape");

    // EDHIP_FLAG_ZERO_GRADIENT: the dense blocks to clear (checked for every input before anything is enqueued)
    const bool zero = gradient && (flags & EDHIP_FLAG_ZERO_GRADIENT);
    char* zero_ptr[EDHIP_MAX_INPUTS];
    long long zero_bytes[EDHIP_MAX_INPUTS];
    if (zero) {
        for (int i = 0; i < ninputs; ++i)
            if (!dense_block(inputs[i], &zero_ptr[i], &zero_bytes[i]))
                return fail(err, errlen, EDHIP_ERR_INVALID, "EDHIP_FLAG_ZERO_GRADIENT needs dense gradient arrays");
    }
    // ---- shared geometry ------------------------------------------------------------------------
    // (a RAW_DISPLACEMENT call filters the control grid here, in one workgroup: the spare workgroups of that launch
    // clear the first gradient block -- cleared[0] -- instead of the tables launch further down)
    GridGeom g;
    bool cleared[EDHIP_MAX_INPUTS] = {};
    // The grid prefilter of a RAW_DISPLACEMENT call is not launched here: the tile path's tables kernel filters the
    // grid inside its own launch (DeformBatch::gridpf); every other route gets the one-workgroup launch in front of
    // it (grid_now: its spare workgroups clear the first gradient block).
    GridPrefilter gpf;
    GridStamp gpf_stamp;
    memset(&gpf, 0, sizeof(gpf));
    {
        int64_t in_len[kMaxAxes], out_len[kMaxAxes];
        for (int k = 0; k < naxis; ++k) {
            in_len[k] = inputs[0].shape[axis[k]];
            out_len[k] = outputs[0].shape[axis[k]];
        }
        bool f64_volumes = false;
        for (int i = 0; i < ninputs; ++i)
            f64_volumes = f64_volumes || inputs[i].dtype == EDHIP_F64;
        const int st = make_geometry(displacement, in_len, out_len, output_offset, naxis, affine, flags,
                                     stream, g, err, errlen, nullptr, 0, nullptr, &gpf, &gpf_stamp, f64_volumes);
        if (st != EDHIP_OK)
            return st;
    }
    auto grid_now = [&]() -> hipError_t {
        if (gpf.total <= 0)
            return hipSuccess;
        if (zero && zero_bytes[0] > 0 && !cleared[0] && ((uintptr_t)zero_ptr[0] & 15) == 0) {
            gpf.zero_ptr = zero_ptr[0];
            gpf.zero_bytes = zero_bytes[0];
            cleared[0] = true;
        }
        const hipError_t e = launch_grid_prefilter(gpf, stream);
        gpf.total = 0;
        if (e == hipSuccess)
            *grid_stamp(stream) = gpf_stamp;
        return e;
    };
    auto clear_now = [&](int i) -> hipError_t {
        if (cleared[i])
            return hipSuccess;
        cleared[i] = true;
        return zero_bytes[i] > 0 ? hipMemsetAsync(zero_ptr[i], 0, (size_t)zero_bytes[i], stream) : hipSuccess;
    };
    if (g.nvox <= 0) {
        if (zero)
            for (int i = 0; i < ninputs; ++i)
                if (clear_now(i) != hipSuccess)
                    return fail(err, errlen, EDHIP_ERR_DEVICE, "clearing the gradient arrays failed");
        return EDHIP_OK;
    }

    // ---- one launch per input/output pair -----------------------------------------------------------
    for (int i = 0; i < ninputs; ++i) {
        const edhip_array& in = inputs[i];
        const edhip_array& out = outputs[i];
        IOView v;
        // float32 volume, 16-bit float on the output side (forward: narrowed at the store; gradient: dY widened at
        // the load), asked for with EDHIP_FLAG_FAST: the level-1 tile kernels take it, or nobody (EDHIP_ERR_UNSUPPORTED,
        // nothing launched; without the flag the exact kernels take the pair like any other).  The view pretends float32 with doubled output strides (IOView::out16).
        const int out16 = ((flags & EDHIP_FLAG_FAST) && !(flags & EDHIP_FLAG_EXACT) && in.dtype == EDHIP_F32)
                              ? (out.dtype == EDHIP_F16 ? 1 : (out.dtype == EDHIP_BF16 ? 2 : 0)) : 0;
        {
            edhip_array out32 = out;
            if (out16) {
                out32.dtype = EDHIP_F32;
                for (int d = 0; d < out.ndim; ++d)
                    out32.stride_bytes[d] = out.stride_bytes[d] * 2;
            }
            const int st = make_view(in, out16 ? out32 : out, naxis, axis + i * naxis, orders[i], modes[i], cvals[i], v,
                                     err, errlen);
            if (st != EDHIP_OK)
                return st;
            v.out16 = out16;
        }
        if (out16 && (v.nsteps <= 0 || ((uintptr_t)out.data & 3) || !deform_fast_supported(g, v, gradient) ||
                      !deform_tile_supported(g, v, gradient != 0)))
            return fail(err, errlen, EDHIP_ERR_UNSUPPORTED, "16-bit output next to a float32 volume: outside the tile kernels");
        if (v.nsteps <= 0) {
            if (zero && clear_now(i) != hipSuccess)
                return fail(err, errlen, EDHIP_ERR_DEVICE, "clearing the gradient arrays failed");
            continue;
        }

        bool use_fast;
        if (flags & EDHIP_FLAG_EXACT)
            use_fast = false;
        else if (flags & EDHIP_FLAG_FAST)
            use_fast = deform_fast_supported(g, v, gradient);
        else
            use_fast = (in.dtype == EDHIP_F32 || in.dtype == EDHIP_F64) && out.dtype == in.dtype &&
                       deform_fast_supported(g, v, gradient);
        // order-0 label maps of any dtype: bit-equal to the exact kernels, so AUTO takes it too
        const bool use_label = !(flags & EDHIP_FLAG_EXACT) && !use_fast && in.dtype != EDHIP_F32 &&
                               in.dtype != EDHIP_F64 && deform_label_supported(g, v, gradient);
        // 8- / 16-bit integer volumes, orders 1-5: bit-equal as well (near-tie voxels redone exactly)
        const bool use_int = !(flags & EDHIP_FLAG_EXACT) && !use_fast && !use_label && deform_int_supported(g, v, gradient);
        hipError_t e = hipSuccess;
        const bool tile = !use_label && !use_int && use_fast && deform_tile_supported(g, v, gradient != 0);
        // (label maps and the integer fast path run the tables kernel of the tile path as well: it filters a raw grid)
        DeformBatch lone;
        lone.nbatch = 1;
        lone.in_bstride = lone.out_bstride = lone.disp_bstride = 0;
        // (the tables launch filters a raw grid in LDS next to its plane slab: only where both fit -- a flat grid of
        // 1342-1365 points would pass 64 KiB with the kernel's static LDS; such a grid gets the one-workgroup launch first)
        const bool gpf_in_tables = gpf.total > 0 && !ed_env("EDHIP_GRIDPF_SEPARATE") &&
                                   8 * (3 * (size_t)g.ncp[1] * (size_t)g.ncp[2] + (size_t)gpf.total) <= 60 * 1024;
        lone.gridpf = (gpf_in_tables && (use_label || use_int)) ? &gpf : nullptr;
        if (!tile && !lone.gridpf)
            e = grid_now();
        if (e == hipSuccess && zero && !tile)
            e = clear_now(i);
        if (e != hipSuccess)
            ;
        else if (use_label || use_int) {
            e = use_label ? launch_deform_label(g, v, stream, &lone) : launch_deform_int(g, v, stream, &lone);
            if (lone.gridpf_done) {
                gpf.total = 0;
                *grid_stamp(stream) = gpf_stamp;
            } else if (lone.gridpf && e == hipSuccess) {
                e = hipErrorUnknown;       // (cannot happen: every route of the label / integer paths starts with the tables launch)
            }
        }
        else if (!use_fast)
            e = launch_deform_exact(g, v, gradient != 0, stream);
        else if (tile) {
            DeformBatch one;
            one.nbatch = 1;
            one.in_bstride = one.out_bstride = one.disp_bstride = 0;
            one.box_mode = (!gradient && (flags & EDHIP_FLAG_KEEP_BOXES)) ? 1
                           : ((gradient && (flags & EDHIP_FLAG_USE_BOXES)) ? 2 : 0);
            one.disp_id = displacement->data;
            one.raw = (flags & EDHIP_FLAG_RAW_DISPLACEMENT) ? 1 : 0;
            one.strong = (flags & EDHIP_FLAG_STRONG_FIELD) ? 1 : 0;
            one.gridpf = gpf_in_tables ? &gpf : nullptr;
            if (!one.gridpf && grid_now() != hipSuccess)
                return fail(err, errlen, EDHIP_ERR_DEVICE, "grid prefilter launch");
            // (several inputs share the geometry: the boxes are those of the last forward launch,
            // which is what a gradient call with the same inputs reads them for)
            // The tile path clears the block inside its tables launch -- when its start is 16-byte aligned.  A
            // block it cannot take is cleared HERE, in front of the scatter (it used to be cleared behind a
            // launch that returned success with zero_done unset: the computed gradient was wiped).
            hipError_t ce = hipSuccess;
            if (zero && !cleared[i]) {
                if (((uintptr_t)zero_ptr[i] & 15) == 0) {
                    one.zero_ptr = zero_ptr[i];
                    one.zero_bytes = zero_bytes[i];
                } else {
                    ce = clear_now(i);
                }
            }
            e = ce == hipSuccess ? launch_deform_tile(g, v, gradient != 0, stream, &one) : ce;
            if (one.gridpf_done) {
                gpf.total = 0;
                *grid_stamp(stream) = gpf_stamp;
            }
            if (out16 && e == hipErrorNotSupported) {
                (void)hipGetLastError();
                return fail(err, errlen, EDHIP_ERR_UNSUPPORTED, "16-bit output next to a float32 volume: outside the level-1 tile kernels");
            }
            if (e == hipErrorNotSupported) {
                // (a wide control grid the level-1 kernels declined -- batches, layouts: nothing was launched, the
                // row kernel takes the call; the gradient block is cleared below if the tile path has not done it)
                (void)hipGetLastError();
                if (grid_now() != hipSuccess)
                    return fail(err, errlen, EDHIP_ERR_DEVICE, "grid prefilter launch");
                if (zero && !cleared[i] && clear_now(i) != hipSuccess)
                    return fail(err, errlen, EDHIP_ERR_DEVICE, "clearing the gradient arrays failed");
                e = launch_deform_fast(g, v, gradient != 0, stream);
            }
            // (a tile path that declined the call -- or served it on a route without the tables' fill -- has
            // not touched the block: cleared now, and whoever takes the call next finds it cleared)
            // (zero_done unset behind an aligned block: the tables launch -- which precedes every scatter of the
            // tile path -- was not made, so nothing has been added to the block yet)
            if (zero && one.zero_ptr && one.zero_done)
                cleared[i] = true;
            if (zero && one.zero_ptr && !one.zero_done && (e == hipSuccess || e == hipErrorNotSupported)) {
                const hipError_t c2 = clear_now(i);
                if (e == hipSuccess)
                    e = c2;
            }
        }
        else
            e = launch_deform_fast(g, v, gradient != 0, stream);
        if (e != hipSuccess)
            return hip_fail(err, errlen, e, "deform kernel launch");
    }
    return EDHIP_OK;
}

int edhip_deform_batch(int gradient, int nbatch, const edhip_array* inputs,
                       const edhip_array* displacements, const int64_t* output_offset,
                       const edhip_array* outputs, int naxis, const int32_t* axis, int32_t order,
                       int32_t mode, double cval, const double* affine, uint32_t flags,
                       void* hip_stream, char* err, size_t errlen)
{
    if (err && errlen)
        err[0] = 0;
    if (nbatch < 0 || (nbatch > 0 && (!inputs || !displacements || !outputs)))
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid batch");
    if (nbatch == 0)
        return EDHIP_OK;          // nothing to do (and no descriptor to read)
    ed::StreamGuard guard((hipStream_t)hip_stream);
    if (gradient && (flags & EDHIP_FLAG_ZERO_GRADIENT)) {
        // batches: cleared up front, item by item (the single-launch path below shares one tables launch)
        for (int b = 0; b < nbatch; ++b) {
            char* zp = nullptr;
            long long zb = 0;
            if (!dense_block(inputs[b], &zp, &zb))
                return fail(err, errlen, EDHIP_ERR_INVALID, "EDHIP_FLAG_ZERO_GRADIENT needs dense gradient arrays");
            if (zb > 0 && hipMemsetAsync(zp, 0, (size_t)zb, (hipStream_t)hip_stream) != hipSuccess)
                return fail(err, errlen, EDHIP_ERR_DEVICE, "clearing the gradient arrays failed");
        }
        flags &= ~(uint32_t)EDHIP_FLAG_ZERO_GRADIENT;
    }
    // ---- one set of launches for the whole batch (the strip index of the tile kernels carries the
    //      sample): same-shaped float volumes at a constant distance, one prefiltered grid each ------
    {
        using namespace ed;
        DeformBatch db;
        db.nbatch = nbatch;
        db.box_mode = (!gradient && (flags & EDHIP_FLAG_KEEP_BOXES)) ? 1
                      : ((gradient && (flags & EDHIP_FLAG_USE_BOXES)) ? 2 : 0);
        db.disp_id = displacements[0].data;
        db.strong = (flags & EDHIP_FLAG_STRONG_FIELD) ? 1 : 0;
        const bool candidate = nbatch >= 2 && naxis == 3 && axis && !(flags & (EDHIP_FLAG_EXACT | EDHIP_FLAG_RAW_DISPLACEMENT)) &&
                               (inputs[0].dtype == EDHIP_F32 || inputs[0].dtype == EDHIP_F64) &&
                               outputs[0].dtype == inputs[0].dtype && inputs[0].ndim == outputs[0].ndim &&
                               inputs[0].ndim >= 3 && inputs[0].ndim <= EDHIP_MAX_DIMS &&
                               displacements[0].ndim == 4 && displacements[0].shape[0] == 3 &&
                               uniform_batch(inputs, nbatch, &db.in_bstride) &&
                               uniform_batch(outputs, nbatch, &db.out_bstride) &&
                               uniform_batch(displacements, nbatch, &db.disp_bstride);
        bool axes_ok = candidate && order >= 0 && order <= 5 && mode >= 0 && mode <= 4;
        for (int j = 0; axes_ok && j < naxis; ++j)
            axes_ok = axis[j] >= 0 && axis[j] < inputs[0].ndim && (j == 0 || axis[j] > axis[j - 1]);
        if (axes_ok) {
            hipStream_t stream = (hipStream_t)hip_stream;
            GridGeom g;
            int64_t in_len[kMaxAxes], out_len[kMaxAxes];
            bool shapes_ok = dtype_ok(displacements[0].dtype);
            for (int k = 0; k < naxis; ++k) {
                in_len[k] = inputs[0].shape[axis[k]];
                out_len[k] = outputs[0].shape[axis[k]];
                shapes_ok = shapes_ok && displacements[0].shape[k + 1] > 0 && in_len[k] >= 2;
            }
            IOView v;
            if (shapes_ok &&
                make_geometry(&displacements[0], in_len, out_len, output_offset, naxis, affine, flags, stream, g,
                              nullptr, 0) == EDHIP_OK &&
                make_view(inputs[0], outputs[0], naxis, axis, order, mode, cval, v, nullptr, 0) == EDHIP_OK &&
                g.nvox > 0 && v.nsteps > 0 && deform_tile_supported(g, v, gradient != 0)) {
                const hipError_t e = launch_deform_tile(g, v, gradient != 0, stream, &db);
                if (e == hipSuccess)
                    return EDHIP_OK;
                if (e != hipErrorNotSupported)
                    return hip_fail(err, errlen, e, "deform kernel launch");
                (void)hipGetLastError();
            }
        }
    }
    // ---- otherwise: item by item (every shape, dtype and flag edhip_deform takes) ---------------------
    for (int b = 0; b < nbatch; ++b) {
        // stream order keeps item b + 1's control grid / tables (which reuse the workspace) behind
        // item b's kernels
        const int st = edhip_deform(gradient, 1, inputs + b, displacements + b, output_offset, outputs + b,
                                    naxis, axis, &order, &mode, &cval, affine, flags, hip_stream, err,
                                    errlen);
        if (st != EDHIP_OK)
            return st;
    }
    return EDHIP_OK;
}

int edhip_deform_batch_strided(int gradient, int nbatch, const edhip_array* input0,
                               int64_t input_batch_stride, const edhip_array* displacement0,
                               int64_t displacement_batch_stride, const int64_t* output_offset,
                               const edhip_array* output0, int64_t output_batch_stride, int naxis,
                               const int32_t* axis, int32_t order, int32_t mode, double cval,
                               const double* affine, uint32_t flags, void* hip_stream, char* err,
                               size_t errlen)
{
    if (err && errlen)
        err[0] = 0;
    if (nbatch < 0 || (nbatch > 0 && (!input0 || !displacement0 || !output0)))
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid batch");
    if (nbatch == 0)
        return EDHIP_OK;
    std::vector<edhip_array> all;
    try {
        all.resize(3 * (size_t)nbatch);
    } catch (...) {
        return fail(err, errlen, EDHIP_ERR_MEMORY, "out of host memory");
    }
    edhip_array* in = all.data();
    edhip_array* disp = in + nbatch;
    edhip_array* out = disp + nbatch;
    for (int b = 0; b < nbatch; ++b) {
        in[b] = *input0;
        in[b].data = (char*)input0->data + (int64_t)b * input_batch_stride;
        disp[b] = *displacement0;
        disp[b].data = (char*)displacement0->data + (int64_t)b * displacement_batch_stride;
        out[b] = *output0;
        out[b].data = (char*)output0->data + (int64_t)b * output_batch_stride;
    }
    return edhip_deform_batch(gradient, nbatch, in, disp, output_offset, out, naxis, axis, order, mode, cval,
                              affine, flags, hip_stream, err, errlen);
}

int edhip_release_scratch(void)
{
    ed::workspace_release_all();
    return EDHIP_OK;
}

int edhip_profile_dominant(int enable)
{
    ed::tile_profile_enable(enable);
    return EDHIP_OK;
}

double edhip_profile_last_us(void) { return ed::tile_profile_last_us(); }

int edhip_source_box(const edhip_array* displacement, const int64_t* in_len, const int64_t* out_len,
                     const int64_t* output_offset, int naxis, const double* affine, uint32_t flags,
                     void* hip_stream, int64_t* box, char* err, size_t errlen)
{
    using namespace ed;
    hipStream_t stream = (hipStream_t)hip_stream;
    StreamGuard guard(stream);
    if (err && errlen)
        err[0] = 0;
    if (!in_len || !out_len || !box || naxis < 1)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis list");
    if (naxis > kMaxAxes)
        return fail(err, errlen, EDHIP_ERR_UNSUPPORTED,
                    "more than %d deformed axes are not supported on the GPU", kMaxAxes);
    if (!displacement || displacement->ndim != naxis + 1 || displacement->shape[0] != naxis)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid displacement shape");
    if (!dtype_ok(displacement->dtype))
        return fail(err, errlen, EDHIP_ERR_DTYPE, "data type not supported");
    for (int k = 0; k <= naxis; ++k)
        if (displacement->shape[k] <= 0)
            return fail(err, errlen, EDHIP_ERR_INVALID, "invalid displacement shape");
    GridGeom g;
    const int st = make_geometry(displacement, in_len, out_len, output_offset, naxis, affine, flags,
                                 stream, g, err, errlen);
    if (st != EDHIP_OK)
        return st;
    hipError_t e = hipSuccess;
    // the box lives in the last bytes of the workspace head (the control grid uses at most 32 KiB)
    char* ws = (char*)workspace_reserve(stream, deform_tile_workspace_bytes(g), &e);
    if (!ws)
        return hip_fail(err, errlen, e, "scratch allocation");
    int* dbox = (int*)(ws + kWorkspaceGridBytes - 64);
    e = launch_source_box(g, dbox, stream, (flags & EDHIP_FLAG_FAST) != 0);
    if (e == hipErrorNotSupported)
        return fail(err, errlen, EDHIP_ERR_UNSUPPORTED,
                    "edhip_source_box: control grids are limited to 7680 values and 4 deformed axes");
    if (e != hipSuccess)
        return hip_fail(err, errlen, e, "source box launch");
    int hbox[2 * kMaxAxes];
    e = hipMemcpyAsync(hbox, dbox, sizeof(int) * 2 * naxis, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(stream);
    if (e != hipSuccess)
        return hip_fail(err, errlen, e, "source box read-back");
    for (int k = 0; k < 2 * naxis; ++k)
        box[k] = hbox[k];
    return EDHIP_OK;
}

// one filter pass; window: device-side window (only the whole-line tile kernels take one: EDHIP_ERR_UNSUPPORTED
// otherwise, nothing launched); dry: the checks of a windowed pass without its launch
static int filter1d_impl(const edhip_array* input, const edhip_array* output, int axis, int order,
                         int transpose, uint32_t flags, void* hip_stream, const int32_t* window, bool dry,
                         char* err, size_t errlen)
{
    using namespace ed;
    hipStream_t stream = (hipStream_t)hip_stream;
    StreamGuard guard(stream);
    if (err && errlen)
        err[0] = 0;
    if (!input || !output)
        return fail(err, errlen, EDHIP_ERR_INVALID, "missing array");
    if (order < 0 || order > 5)                                   // _deform_grid.c:71-74
        return fail(err, errlen, EDHIP_ERR_INVALID, "spline order not supported");
    if (input->ndim < 1 || input->ndim > EDHIP_MAX_DIMS || input->ndim != output->ndim)
        return fail(err, errlen, EDHIP_ERR_INVALID, "input and output dimensions should match");
    if (axis < 0)
        axis += input->ndim;                                      // _deform_grid.c:75-77
    if (axis < 0 || axis >= input->ndim)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis");
    if (!dtype_ok(input->dtype) || !dtype_ok(output->dtype))
        return fail(err, errlen, EDHIP_ERR_DTYPE, "data type not supported");
    for (int d = 0; d < input->ndim; ++d)
        if (input->shape[d] != output->shape[d])
            return fail(err, errlen, EDHIP_ERR_INVALID, "input and output shapes should match");

    FilterParams p;
    memset(&p, 0, sizeof(p));
    p.in = (const char*)input->data;
    p.out = (char*)output->data;
    p.in_dtype = input->dtype;
    p.out_dtype = output->dtype;
    p.transpose = transpose != 0;
    p.len = input->shape[axis];
    p.in_axis_stride = input->stride_bytes[axis];
    p.out_axis_stride = output->stride_bytes[axis];
    p.nlines = 1;
    for (int d = 0; d < input->ndim; ++d) {
        if (d == axis)
            continue;
        p.outer_len[p.nouter] = input->shape[d];
        p.in_outer_stride[p.nouter] = input->stride_bytes[d];
        p.out_outer_stride[p.nouter] = output->stride_bytes[d];
        p.nlines *= input->shape[d];
        p.nouter++;
    }
    if (p.len <= 0 || p.nlines <= 0)
        return EDHIP_OK;

    // poles.  transpose: the sqrt() expressions of deform.c:1063-1084; forward: SciPy's correctly
    // rounded literals (they differ from the expressions by an ulp or so -- see oracle/ed_oracle.c)
    switch (order) {
    case 2:
        p.npoles = 1;
        p.pole[0] = p.transpose ? std::sqrt(8.0) - 3.0 : -0.171572875253809902396622551580603843;
        break;
    case 3:
        p.npoles = 1;
        p.pole[0] = p.transpose ? std::sqrt(3.0) - 2.0 : -0.267949192431122706472553658494127633;
        break;
    case 4:
        p.npoles = 2;
        if (p.transpose) {
            p.pole[0] = std::sqrt(664.0 - std::sqrt(438976.0)) + std::sqrt(304.0) - 19.0;
            p.pole[1] = std::sqrt(664.0 + std::sqrt(438976.0)) - std::sqrt(304.0) - 19.0;
        } else {
            p.pole[0] = -0.361341225900220177092212841325675255;
            p.pole[1] = -0.013725429297339121360331226939128204;
        }
        break;
    case 5:
        p.npoles = 2;
        if (p.transpose) {
            p.pole[0] = std::sqrt(67.5 - std::sqrt(4436.25)) + std::sqrt(26.25) - 6.5;
            p.pole[1] = std::sqrt(67.5 + std::sqrt(4436.25)) - std::sqrt(26.25) - 6.5;
        } else {
            p.pole[0] = -0.430575347099973791851434783493520110;
            p.pole[1] = -0.043096288203264653822712376822550182;
        }
        break;
    default: p.npoles = 0; break;
    }
    p.gain = 1.0;
    for (int h = 0; h < p.npoles; ++h) {
        p.gain *= (1.0 - p.pole[h]) * (1.0 - 1.0 / p.pole[h]);          // deform.c:1086-1088
        p.pole_pow[h] = std::pow(p.pole[h], (double)(p.len - 1));       // host libm, like the reference
        const int max = (int)std::ceil(std::log(1e-15) / std::log(std::fabs(p.pole[h])));
        p.trunc_branch[h] = max < p.len;                                // deform.c:1119,1134
    }

    // fast path for float32 / float64 unless the caller asks for the exact one; same operator, no
    // scratch, ~1e-16 relative to the sequential recursion (see spline_fast.hip)
    // (16-bit float storage on one side of a float32 pass: the whole-line tile kernels widen / narrow it on the way)
    const bool half_in = (input->dtype == EDHIP_F16 || input->dtype == EDHIP_BF16) && output->dtype == EDHIP_F32;
    const bool want_fast = !(flags & EDHIP_FLAG_EXACT) &&
                           (input->dtype == EDHIP_F32 || input->dtype == EDHIP_F64 || half_in);
    if (want_fast && p.npoles >= 1) {
        const hipError_t e = launch_spline_filter_fast(p, order, input->ndim, axis, input->shape,
                                                       input->stride_bytes, output->stride_bytes,
                                                       stream, window, dry);
        if (e == hipSuccess)
            return EDHIP_OK;
        if (e != hipErrorNotSupported)
            return hip_fail(err, errlen, e, "spline filter launch");
        (void)hipGetLastError();
    }
    if (window || dry)
        return fail(err, errlen, EDHIP_ERR_UNSUPPORTED, "windowed prefilter: outside the whole-line tile kernels");
    // a float32 pass with 16-bit float storage on one side, asked for with EDHIP_FLAG_FAST: the tile kernels or nobody
    // (without the flag the exact kernel takes the pair like any other)
    const bool half_out = input->dtype == EDHIP_F32 && (output->dtype == EDHIP_F16 || output->dtype == EDHIP_BF16);
    if ((half_in || half_out) && (flags & EDHIP_FLAG_FAST) && !(flags & EDHIP_FLAG_EXACT))
        return fail(err, errlen, EDHIP_ERR_UNSUPPORTED, "16-bit storage next to a float32 pass: outside the whole-line tile kernels");

    const bool need_ws = p.npoles > 0 && p.len >= 2;
    if (need_ws) {
        // bound the fp64 scratch at 256 MiB; chunks of lines reuse it in stream order
        const int64_t budget = (int64_t)256 << 20;
        int64_t lines = budget / 8 / p.len;
        if (lines < 256)
            lines = 256;
        lines -= lines % 256;
        if (lines > p.nlines)
            lines = p.nlines;
        p.ws_lines = lines;
        hipError_t e = hipSuccess;
        void* ws = workspace_reserve(stream, kWorkspaceGridBytes + (size_t)lines * (size_t)p.len * 8, &e);
        if (!ws)
            return hip_fail(err, errlen, e, "scratch allocation");
        p.ws = (double*)((char*)ws + kWorkspaceGridBytes);
        e = launch_spline_filter(p, stream);
        if (e != hipSuccess)
            return hip_fail(err, errlen, e, "spline filter launch");
    } else {
        p.ws = nullptr;
        p.ws_lines = p.nlines;
        const hipError_t e = launch_spline_filter(p, stream);
        if (e != hipSuccess)
            return hip_fail(err, errlen, e, "spline filter launch");
    }
    return EDHIP_OK;
}

int edhip_spline_filter1d(const edhip_array* input, const edhip_array* output, int axis, int order,
                          int transpose, uint32_t flags, void* hip_stream, char* err, size_t errlen)
{
    return filter1d_impl(input, output, axis, order, transpose, flags, hip_stream, nullptr, false, err, errlen);
}

int edhip_spline_filter_axes_window(const edhip_array* input, const edhip_array* output, int naxes,
                                    const int32_t* axes, int order, int transpose, const int32_t* window,
                                    uint32_t flags, void* hip_stream, char* err, size_t errlen)
{
    if (err && errlen)
        err[0] = 0;
    if (naxes < 0 || (naxes > 0 && !axes) || !window)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis list / window");
    ed::StreamGuard guard((hipStream_t)hip_stream);
    // every pass must be one the tile kernels take, BEFORE the first one is launched
    for (int i = 0; i < naxes; ++i) {
        const int st = filter1d_impl(i == 0 ? input : output, output, axes[i], order, transpose, flags, hip_stream,
                                     window, true, err, errlen);
        if (st != EDHIP_OK)
            return st;
    }
    for (int i = 0; i < naxes; ++i) {
        const int st = filter1d_impl(i == 0 ? input : output, output, axes[i], order, transpose, flags, hip_stream,
                                     window, false, err, errlen);
        if (st != EDHIP_OK)
            return st;
    }
    return EDHIP_OK;
}

int edhip_source_window(const edhip_array* displacement, const int64_t* in_len, const int64_t* out_len,
                        const int64_t* output_offset, int naxis, const double* affine, int ndim,
                        const int64_t* shape, const int32_t* axis, int order, int mode, int margin, int align,
                        int minlen, uint32_t flags, void* hip_stream, int32_t* window, char* err, size_t errlen)
{
    using namespace ed;
    hipStream_t stream = (hipStream_t)hip_stream;
    StreamGuard guard(stream);
    if (err && errlen)
        err[0] = 0;
    if (!in_len || !out_len || !window || !shape || !axis || naxis < 1 || ndim < naxis || ndim > EDHIP_MAX_DIMS)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis list");
    if (naxis > 4)
        return fail(err, errlen, EDHIP_ERR_UNSUPPORTED, "edhip_source_window: up to 4 deformed axes");
    if (!displacement || displacement->ndim != naxis + 1 || displacement->shape[0] != naxis)
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid displacement shape");
    if (!dtype_ok(displacement->dtype))
        return fail(err, errlen, EDHIP_ERR_DTYPE, "data type not supported");
    int64_t points = naxis;
    for (int k = 0; k <= naxis; ++k) {
        if (displacement->shape[k] <= 0)
            return fail(err, errlen, EDHIP_ERR_INVALID, "invalid displacement shape");
        if (k > 0)
            points *= displacement->shape[k];
    }
    if (points > 7680)
        return fail(err, errlen, EDHIP_ERR_UNSUPPORTED, "edhip_source_window: control grids are limited to 7680 values");
    SourceWindow sw{};
    sw.out = window;
    sw.ndim = ndim;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] <= 0 || shape[d] > 0x3fffffff)
            return fail(err, errlen, EDHIP_ERR_INVALID, "invalid shape");
        sw.shape[d] = (int)shape[d];
    }
    for (int k = 0; k < naxis; ++k) {
        if (axis[k] < 0 || axis[k] >= ndim || shape[axis[k]] != in_len[k])
            return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis list");
        sw.axis[k] = axis[k];
    }
    sw.order = order;
    sw.mode = mode;
    sw.margin = margin < 0 ? 0 : margin;
    sw.align = align < 1 ? 1 : align;
    sw.minlen = minlen < 1 ? 1 : minlen;
    GridGeom g;
    const int st = make_geometry(displacement, in_len, out_len, output_offset, naxis, affine, flags,
                                 stream, g, err, errlen);
    if (st != EDHIP_OK)
        return st;
    hipError_t e = hipSuccess;
    char* ws = (char*)workspace_reserve(stream, deform_tile_workspace_bytes(g), &e);
    if (!ws)
        return hip_fail(err, errlen, e, "scratch allocation");
    int* dbox = (int*)(ws + kWorkspaceGridBytes - 64);
    e = launch_source_box(g, dbox, stream, true, &sw);
    if (e != hipSuccess)
        return hip_fail(err, errlen, e, "source window launch");
    return EDHIP_OK;
}

int edhip_spline_filter_axes(const edhip_array* input, const edhip_array* output, int naxes,
                             const int32_t* axes, int order, int transpose, uint32_t flags,
                             void* hip_stream, char* err, size_t errlen)
{
    if (err && errlen)
        err[0] = 0;
    if (naxes < 0 || (naxes > 0 && !axes))
        return fail(err, errlen, EDHIP_ERR_INVALID, "invalid axis list");
    ed::StreamGuard guard((hipStream_t)hip_stream);
    const bool scratch_in = (flags & EDHIP_FLAG_SCRATCH_INPUT) && naxes > 1;
    if (naxes > 0 && input && output && input->dtype != output->dtype && (flags & EDHIP_FLAG_FAST) &&
        !(flags & EDHIP_FLAG_EXACT)) {
        // the one pass that converts (first of a widening chain, last of a narrowing one): can the tile kernels take
        // it?  Decided before anything is launched.
        const int i = (flags & EDHIP_FLAG_SCRATCH_INPUT) ? naxes - 1 : 0;
        const int st = filter1d_impl(input, output, axes[i], order, transpose, flags, hip_stream, nullptr, true, err, errlen);
        if (st != EDHIP_OK)
            return st;
    }
    for (int i = 0; i < naxes; ++i) {
        // the reference's loop (deform_grid.py:157-162, :279-284): input -> output, then in place;
        // EDHIP_FLAG_SCRATCH_INPUT: in place on the input, the last pass input -> output
        const edhip_array* src = scratch_in ? input : (i == 0 ? input : output);
        const edhip_array* dst = scratch_in ? (i == naxes - 1 ? output : input) : output;
        const int st = edhip_spline_filter1d(src, dst, axes[i], order, transpose, flags, hip_stream, err, errlen);
        if (st != EDHIP_OK)
            return st;
    }
    return EDHIP_OK;
}

}  // extern "C"
