// spline_filter.hip -- K3 (B-spline prefilter, mirror boundary) and K4 (its exact transpose).
//
// K3 restates scipy.ndimage.spline_filter1d (third-party; call sites deform_grid.py:160,168,271;
// algorithm in SURVEY.md Appendix A step 12, pinned bit-for-bit against SciPy 1.15.3 by the
// oracle tests).  K4 restates NI_SplineFilter1DGrad's per-line recursion, deform.c:1116-1156.
//
// Compiled with -ffp-contract=off: every line is filtered in fp64 in the reference's operation
// order and rounded to the output dtype once, like the reference's double line buffer
// (deform.c:1106-1162, from_nd_image.c:341-350,422-431).  The result is bit-identical to the CPU
// path for every dtype.
//
// Mapping: one thread per line (a first-order IIR is sequential along the line; all parallelism
// is across lines).  The fp64 working copy of the lines lives in a line-interleaved scratch
// buffer ws[i * nlines + line], so that the 64 lanes of a wave, which hold 64 different lines,
// always touch 64 consecutive doubles -- coalesced whatever the filtered axis is.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "ed_device.h"
#include "ed_params.h"
#include "ed_gridfilter.h"

namespace ed {

namespace {

struct LineAddr {
    const char* in;
    char* out;
};

__device__ __forceinline__ LineAddr line_address(const FilterParams& p, int64_t line)
{
    int64_t in_off = 0, out_off = 0, r = line;
    for (int d = p.nouter - 1; d >= 0; --d) {
        const int64_t q = r / p.outer_len[d];
        const int64_t c = r - q * p.outer_len[d];
        in_off += c * p.in_outer_stride[d];
        out_off += c * p.out_outer_stride[d];
        r = q;
    }
    return {p.in + in_off, p.out + out_off};
}

// forward prefilter of one line in place (SciPy: [gain applied by the caller], then per pole causal
// init / causal recursion / anti-causal init / anti-causal recursion); element i at ws[i * wst]
__device__ __forceinline__ void forward_line(double* ws, const int64_t wst, const int64_t n, const FilterParams& p)
{
    for (int h = 0; h < p.npoles; ++h) {
        const double z = p.pole[h];
        const double zn1 = p.pole_pow[h];
        // exact mirror initialisation of the causal filter
        double c0 = ws[0] + zn1 * ws[(n - 1) * wst];
        double zi = z;
        for (int64_t i = 1; i < n - 1; ++i) {
            c0 += zi * (ws[i * wst] + zn1 * ws[(n - 1 - i) * wst]);
            zi *= z;
        }
        c0 /= 1 - zn1 * zn1;
        ws[0] = c0;
        // causal recursion c[i] += z c[i-1]; keep the last two values for the anti-causal init
        double prev = c0, prev2 = c0;
        for (int64_t i = 1; i < n; ++i) {
            const double c = ws[i * wst] + z * prev;
            ws[i * wst] = c;
            prev2 = prev;
            prev = c;
        }
        // anti-causal initialisation and recursion c[i] = z (c[i+1] - c[i])
        double next = (z * prev2 + prev) * z / (z * z - 1);
        ws[(n - 1) * wst] = next;
        for (int64_t i = n - 2; i >= 0; --i) {
            const double c = z * (next - ws[i * wst]);
            ws[i * wst] = c;
            next = c;
        }
    }
}

// transpose of the prefilter on one line in place, gain included -- deform.c:1116-1156
__device__ __forceinline__ void transpose_line(double* ws, const int64_t wst, const int64_t len, const FilterParams& p)
{
    for (int h = 0; h < p.npoles; ++h) {
        const double q = p.pole[h];
        const bool last = h == p.npoles - 1;
        const double gl = last ? p.gain : 1.0;      // (x * 1.0 is exact: the intermediate poles are unchanged)
        // adjoint of the anti-causal recursion, running sum for the adjoint of its initialisation
        double x0 = ws[0];
        double sum = q * x0;
        double prev = -q * x0;
        ws[0] = prev;
        for (int64_t ll = 1; ll < len - 1; ++ll) {
            const double x = ws[ll * wst];
            sum = q * (sum + x);
            prev = q * (prev - x);
            ws[ll * wst] = prev;
        }
        sum = (q / (q * q - 1.0)) * (sum + ws[(len - 1) * wst]);
        double up = ws[(len - 2) * wst] + q * sum;     // ln[len-2] += p*sum
        ws[(len - 1) * wst] = sum;                     // ln[len-1]  = sum
        // adjoint of the causal recursion: ln[ll] += p * ln[ll+1], ll = len-2 .. 0
        double next = sum;
        for (int64_t ll = len - 2; ll >= 0; --ll) {
            const double cur = (ll == len - 2 ? up : ws[ll * wst]) + q * next;
            ws[ll * wst] = cur;
            next = cur;
        }
        // adjoint of the causal initial sum (next == ln[0] here)
        if (p.trunc_branch[h]) {
            const double l0 = next;
            double zn = q;
            ws[0] = last ? l0 * p.gain : l0;
            for (int64_t ll = 1; ll < len; ++ll) {
                const double c = ws[ll * wst] + zn * l0;
                ws[ll * wst] = last ? c * p.gain : c;
                zn *= q;
            }
        } else {
            double zn = q;
            const double iz = 1.0 / q;
            double z2n = p.pole_pow[h];
            const double l0 = next / (1.0 - z2n * z2n);
            const double tail = ws[(len - 1) * wst] + z2n * l0;
            z2n *= z2n * iz;
            ws[0] = last ? l0 * p.gain : l0;
            ws[(len - 1) * wst] = last ? tail * p.gain : tail;
            for (int64_t ll = 1; ll <= len - 2; ++ll) {
                const double c = ws[ll * wst] + (zn + z2n) * l0;
                ws[ll * wst] = last ? c * p.gain : c;
                zn *= q;
                z2n *= iz;
            }
        }
        (void)gl;
    }
}

// The same two recursions for a line in an LDS tile (element i at ws[i * kTilePitch]), blocked by 8: the
// 8 LDS reads of a block are issued together and the dependent fp64 chain runs on registers.  With one
// wave per CU nothing else hides the LDS latency, and the compiler does not pipeline the rolled loops
// above.  Same operations on the same operands in the same order: same bits.
constexpr int kBlk = 8;      // (row pitch W of a tile = lines per tile + 1 doubles: odd, conflict-free)
template <int W>
__device__ __forceinline__ void forward_line_tile(double* ws, const int n, const FilterParams& p)
{
    for (int h = 0; h < p.npoles; ++h) {
        const double z = p.pole[h];
        const double zn1 = p.pole_pow[h];
        double c0 = ws[0] + zn1 * ws[(n - 1) * W];
        double zi = z;
        int i = 1;
        for (; i + kBlk <= n - 1; i += kBlk) {
            double a[kBlk], b[kBlk];
#pragma unroll
            for (int k = 0; k < kBlk; ++k) {
                a[k] = ws[(i + k) * W];
                b[k] = ws[(n - 1 - i - k) * W];
            }
#pragma unroll
            for (int k = 0; k < kBlk; ++k) {
                c0 += zi * (a[k] + zn1 * b[k]);
                zi *= z;
            }
        }
        for (; i < n - 1; ++i) {
            c0 += zi * (ws[i * W] + zn1 * ws[(n - 1 - i) * W]);
            zi *= z;
        }
        c0 /= 1 - zn1 * zn1;
        ws[0] = c0;
        double prev = c0, prev2 = c0;
        i = 1;
        for (; i + kBlk <= n; i += kBlk) {
            double a[kBlk];
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                a[k] = ws[(i + k) * W];
#pragma unroll
            for (int k = 0; k < kBlk; ++k) {
                const double c = a[k] + z * prev;
                a[k] = c;
                prev2 = prev;
                prev = c;
            }
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                ws[(i + k) * W] = a[k];
        }
        for (; i < n; ++i) {
            const double c = ws[i * W] + z * prev;
            ws[i * W] = c;
            prev2 = prev;
            prev = c;
        }
        double next = (z * prev2 + prev) * z / (z * z - 1);
        ws[(n - 1) * W] = next;
        i = n - 2;
        for (; i - kBlk + 1 >= 0; i -= kBlk) {
            double a[kBlk];
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                a[k] = ws[(i - k) * W];
#pragma unroll
            for (int k = 0; k < kBlk; ++k) {
                const double c = z * (next - a[k]);
                a[k] = c;
                next = c;
            }
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                ws[(i - k) * W] = a[k];
        }
        for (; i >= 0; --i) {
            const double c = z * (next - ws[i * W]);
            ws[i * W] = c;
            next = c;
        }
    }
}

// transpose_line for a line in an LDS tile, blocked like forward_line_tile (same operations, same order)
template <int W>
__device__ __forceinline__ void transpose_line_tile(double* ws, const int len, const FilterParams& p)
{
    for (int h = 0; h < p.npoles; ++h) {
        const double q = p.pole[h];
        const bool last = h == p.npoles - 1;
        const double x0 = ws[0];
        double sum = q * x0;
        double prev = -q * x0;
        ws[0] = prev;
        int ll = 1;
        for (; ll + kBlk <= len - 1; ll += kBlk) {
            double a[kBlk];
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                a[k] = ws[(ll + k) * W];
#pragma unroll
            for (int k = 0; k < kBlk; ++k) {
                const double x = a[k];
                sum = q * (sum + x);
                prev = q * (prev - x);
                a[k] = prev;
            }
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                ws[(ll + k) * W] = a[k];
        }
        for (; ll < len - 1; ++ll) {
            const double x = ws[ll * W];
            sum = q * (sum + x);
            prev = q * (prev - x);
            ws[ll * W] = prev;
        }
        sum = (q / (q * q - 1.0)) * (sum + ws[(len - 1) * W]);
        const double up = ws[(len - 2) * W] + q * sum;
        ws[(len - 1) * W] = sum;
        double next = up + q * sum;
        ws[(len - 2) * W] = next;
        ll = len - 3;
        for (; ll - kBlk + 1 >= 0; ll -= kBlk) {
            double a[kBlk];
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                a[k] = ws[(ll - k) * W];
#pragma unroll
            for (int k = 0; k < kBlk; ++k) {
                const double cur = a[k] + q * next;
                a[k] = cur;
                next = cur;
            }
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                ws[(ll - k) * W] = a[k];
        }
        for (; ll >= 0; --ll) {
            const double cur = ws[ll * W] + q * next;
            ws[ll * W] = cur;
            next = cur;
        }
        const double g = p.gain;
        if (p.trunc_branch[h]) {
            const double l0 = next;
            double zn = q;
            ws[0] = last ? l0 * g : l0;
            ll = 1;
            for (; ll + kBlk <= len; ll += kBlk) {
                double a[kBlk];
#pragma unroll
                for (int k = 0; k < kBlk; ++k)
                    a[k] = ws[(ll + k) * W];
#pragma unroll
                for (int k = 0; k < kBlk; ++k) {
                    const double c = a[k] + zn * l0;
                    a[k] = last ? c * g : c;
                    zn *= q;
                }
#pragma unroll
                for (int k = 0; k < kBlk; ++k)
                    ws[(ll + k) * W] = a[k];
            }
            for (; ll < len; ++ll) {
                const double c = ws[ll * W] + zn * l0;
                ws[ll * W] = last ? c * g : c;
                zn *= q;
            }
        } else {
            double zn = q;
            const double iz = 1.0 / q;
            double z2n = p.pole_pow[h];
            const double l0 = next / (1.0 - z2n * z2n);
            const double tail = ws[(len - 1) * W] + z2n * l0;
            z2n *= z2n * iz;
            ws[0] = last ? l0 * g : l0;
            ws[(len - 1) * W] = last ? tail * g : tail;
            ll = 1;
            for (; ll + kBlk <= len - 1; ll += kBlk) {
                double a[kBlk];
#pragma unroll
                for (int k = 0; k < kBlk; ++k)
                    a[k] = ws[(ll + k) * W];
#pragma unroll
                for (int k = 0; k < kBlk; ++k) {
                    const double c = a[k] + (zn + z2n) * l0;
                    a[k] = last ? c * g : c;
                    zn *= q;
                    z2n *= iz;
                }
#pragma unroll
                for (int k = 0; k < kBlk; ++k)
                    ws[(ll + k) * W] = a[k];
            }
            for (; ll <= len - 2; ++ll) {
                const double c = ws[ll * W] + (zn + z2n) * l0;
                ws[ll * W] = last ? c * g : c;
                zn *= q;
                z2n *= iz;
            }
        }
    }
}

// One thread per line; LDSWS: the fp64 working copy of the block's lines sits in LDS ([len][blockDim]
// doubles, line-interleaved) instead of the global scratch buffer: short lines (small volumes)
// otherwise pay a dependent global round trip per sample -- 28 us per pass for a 32^3 volume, more
// than the deformation itself.  Same operations in the same order: same bits.
template <bool LDSWS, bool TRANSPOSE>
__global__ __launch_bounds__(256) void prefilter_kernel(const FilterParams p, const int64_t line0,
                                                        const int64_t nl)
{
    extern __shared__ double lws[];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nl)
        return;
    const LineAddr a = line_address(p, line0 + j);
    const int64_t n = p.len;
    const int64_t wst = LDSWS ? (int64_t)blockDim.x : nl;
    double* ws = LDSWS ? lws + threadIdx.x : p.ws + j;      // element i at ws[i * wst]
    if (n < 2 || p.npoles == 0) {
        for (int64_t i = 0; i < n; ++i)
            store_cast(a.out + i * p.out_axis_stride, p.out_dtype,
                       load_as_double(a.in + i * p.in_axis_stride, p.in_dtype));
        return;
    }
    if (TRANSPOSE) {
        for (int64_t i = 0; i < n; ++i)
            ws[i * wst] = load_as_double(a.in + i * p.in_axis_stride, p.in_dtype);
        transpose_line(ws, wst, n, p);
    } else {
        for (int64_t i = 0; i < n; ++i)
            ws[i * wst] = load_as_double(a.in + i * p.in_axis_stride, p.in_dtype) * p.gain;
        forward_line(ws, wst, n, p);
    }
    for (int64_t i = 0; i < n; ++i)
        store_cast(a.out + i * p.out_axis_stride, p.out_dtype, ws[i * wst]);
}

// Lines of 129 .. 313 samples (256^3 volumes): a tile of 64 lines in LDS (fp64, [len][65]: one wave per
// CU owns the 160 KiB).  All 256 threads move the tile -- along the filtered axis when that is the
// contiguous one, across the 64 lines otherwise, so that a wave instruction touches consecutive
// addresses either way -- and one wave runs the 64 recursions.  The per-thread kernel above walks its
// line sample by sample through global memory: a dependent round trip per sample and, for the
// contiguous axis, 64 cache lines per wave instruction (256^3 int16: 340 us per axis pass).
// calls f(std::integral_constant<int, DT>) for the runtime dtype code: the per-element switch of
// load_as_double / store_cast folds away inside f, and its loads can be issued back to back
template <typename F>
__device__ __forceinline__ void with_dtype(int dt, F&& f)
{
#define ED_DT_CASE(D) case D: f(std::integral_constant<int, D>()); break;
    switch (dt) {
    ED_DT_CASE(EDHIP_BOOL) ED_DT_CASE(EDHIP_U8) ED_DT_CASE(EDHIP_I8) ED_DT_CASE(EDHIP_U16) ED_DT_CASE(EDHIP_I16)
    ED_DT_CASE(EDHIP_U32) ED_DT_CASE(EDHIP_I32) ED_DT_CASE(EDHIP_U64) ED_DT_CASE(EDHIP_I64) ED_DT_CASE(EDHIP_F16)
    ED_DT_CASE(EDHIP_BF16) ED_DT_CASE(EDHIP_F32)
    default: f(std::integral_constant<int, EDHIP_F64>()); break;
    }
#undef ED_DT_CASE
}

template <int kTileLines, bool TRANSPOSE>
__global__ __launch_bounds__(256) void prefilter_line_tile_kernel(const FilterParams p)
{
    constexpr int kTilePitch = kTileLines + 1;
    extern __shared__ double lws[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t line0 = (int64_t)blockIdx.x * kTileLines;
    const int nl = (int)(p.nlines - line0 < kTileLines ? p.nlines - line0 : kTileLines);
    const int n = (int)p.len;
    // (no static LDS: the dynamic allocation may then take all 160 KiB)
    int64_t* s_in = reinterpret_cast<int64_t*>(lws + (size_t)n * kTilePitch);
    int64_t* s_out = s_in + kTileLines;
    if (tid < kTileLines) {
        const LineAddr a = line_address(p, line0 + (tid < nl ? tid : nl - 1));
        s_in[tid] = a.in - p.in;
        s_out[tid] = a.out - p.out;
    }
    __syncthreads();
    const bool along_in = p.in_axis_stride == dtype_size(p.in_dtype);
    const bool along_out = p.out_axis_stride == dtype_size(p.out_dtype);
    const double g = TRANSPOSE ? 1.0 : p.gain;
    // ---- tile -> LDS: a wave instruction touches consecutive addresses either along the filtered axis
    //      (lanes <-> samples, waves <-> lines) or across the lines (lanes <-> lines, waves <-> samples)
    with_dtype(p.in_dtype, [&](auto dt) {
        constexpr int DT = decltype(dt)::value;
        if (along_in) {
            for (int l = wave; l < nl; l += 4) {
                const char* base = p.in + s_in[l];
#pragma unroll 4
                for (int i = lane; i < n; i += 64) {
                    const double v = load_as_double(base + (int64_t)i * p.in_axis_stride, DT);
                    lws[i * kTilePitch + l] = TRANSPOSE ? v : v * g;
                }
            }
        } else {
            // lanes <-> lines; tiles of fewer than 64 lines put 64 / kTileLines samples in one wave instruction
            constexpr int SPW = 64 / kTileLines;
            const int l = lane % kTileLines, sub = lane / kTileLines;
            if (l < nl) {
                const char* base = p.in + s_in[l];
#pragma unroll 8
                for (int i = wave * SPW + sub; i < n; i += 4 * SPW) {
                    const double v = load_as_double(base + (int64_t)i * p.in_axis_stride, DT);
                    lws[i * kTilePitch + l] = TRANSPOSE ? v : v * g;
                }
            }
        }
    });
    __syncthreads();
    if (tid < nl) {
        if (TRANSPOSE)
            transpose_line_tile<kTilePitch>(lws + tid, n, p);
        else
            forward_line_tile<kTilePitch>(lws + tid, n, p);
    }
    __syncthreads();
    with_dtype(p.out_dtype, [&](auto dt) {
        constexpr int DT = decltype(dt)::value;
        if (along_out) {
            for (int l = wave; l < nl; l += 4) {
                char* base = p.out + s_out[l];
#pragma unroll 4
                for (int i = lane; i < n; i += 64)
                    store_cast(base + (int64_t)i * p.out_axis_stride, DT, lws[i * kTilePitch + l]);
            }
        } else {
            constexpr int SPW = 64 / kTileLines;
            const int l = lane % kTileLines, sub = lane / kTileLines;
            if (l < nl) {
                char* base = p.out + s_out[l];
#pragma unroll 8
                for (int i = wave * SPW + sub; i < n; i += 4 * SPW)
                    store_cast(base + (int64_t)i * p.out_axis_stride, DT, lws[i * kTilePitch + l]);
            }
        }
    });
}

// Whole-grid order-3 prefilter of a small control grid (<= 4096 points) in ONE launch: the grid
// sits in LDS as doubles, every grid axis is filtered in turn by the same sequential recursion
// as prefilter_kernel (one thread per line), and after each axis the values are rounded to the
// grid's storage dtype exactly like the reference's per-axis `output=displacement_f` round trip
// (deform_grid.py:166-169).  Output: contiguous [naxis][ncp...] array of the same dtype.
__global__ __launch_bounds__(256) void grid_prefilter_kernel(const GridPrefilter p)
{
    __shared__ double s[4096];
    const int tid = threadIdx.x;
    if (blockIdx.x > 0) {
        // the gradient's accumulators (see GridPrefilter::zero_ptr): 64 MB in the ~7 us this launch takes anyway,
        // and a few more -- it used to ride on the tables launch behind this one (25 us for an 8 us kernel)
        const long long nfill = (long long)(gridDim.x - 1) * 256;
        const long long me = (long long)(blockIdx.x - 1) * 256 + tid;
        const long long n16 = p.zero_bytes >> 4;
        int4* p16 = reinterpret_cast<int4*>(p.zero_ptr);
        for (long long i = me; i < n16; i += nfill)
            p16[i] = make_int4(0, 0, 0, 0);
        if (me < (p.zero_bytes & 15))
            p.zero_ptr[(n16 << 4) + me] = 0;
        return;
    }
    const int total = p.total;
    grid_prefilter_in_lds<256>(p, s, tid);
    const int esz = p.elem_size;
    for (int e = tid; e < total; e += 256)
        store_cast(p.out + (int64_t)e * esz, p.dtype, s[e]);
}

}  // namespace

hipError_t launch_grid_prefilter(const GridPrefilter& p, hipStream_t stream)
{
    long long fill = 0;
    if (p.zero_ptr && p.zero_bytes > 0) {
        fill = (p.zero_bytes + 65535) / 65536;          // 64 KiB per workgroup and round
        fill = fill < 1 ? 1 : (fill > 2048 ? 2048 : fill);
    }
    hipLaunchKernelGGL(grid_prefilter_kernel, dim3(1 + (unsigned)fill), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_spline_filter(const FilterParams& p, hipStream_t stream)
{
    if (p.nlines <= 0 || p.len <= 0)
        return hipSuccess;
    static const bool no_ldsws = ed_env("EDHIP_FILTER_NO_LDSWS") != nullptr;      // A/B switch
    static const bool no_short_tile = ed_env("EDHIP_FILTER_NO_SHORT_TILE") != nullptr;
    // Lines of 2 .. 4000 samples: tiles of 64 / 16 / 4 lines in LDS (fp64, [len][lines + 1]), moved by the
    // whole workgroup; one lane per line runs the recursion.  64 lines fit up to 313 samples (the lines of a
    // 256^3 volume), 16 up to 1200, 4 up to 4000 (2-D images: a 1024^2 uint8 image took 0.85 ms per pass on
    // the one-thread-per-line kernel below, its samples walked one dependent global round trip at a time).
    // (Short lines used to go to the one-thread-per-line LDS kernel: 21 us per pass for a 32^3 volume.)
    if (p.len >= 2 && p.npoles > 0 && !no_ldsws && (p.len > 128 || !no_short_tile) && dtype_size(p.in_dtype) > 0 &&
        dtype_size(p.out_dtype) > 0) {
        const size_t budget = 160 * 1024 - 256;
        auto lds_for = [&](int lines) {
            return (size_t)p.len * (lines + 1) * sizeof(double) + 2 * (size_t)lines * sizeof(int64_t);
        };
        const int lines = lds_for(64) <= budget ? 64 : (lds_for(16) <= budget ? 16 : (lds_for(4) <= budget ? 4 : 0));
        const int64_t nblk = lines ? (p.nlines + lines - 1) / lines : 0;
        if (lines && nblk <= 0x7fffffffLL) {
            const size_t lds = lds_for(lines);
            hipError_t e = hipSuccess;
            auto go = [&](auto kernel, int slot) {
                if (lds > 64 * 1024) {
                    // (per device and kernel: a process may drive several GPUs)
                    static std::atomic<unsigned long long> allowed[6];
                    int dev = 0;
                    (void)hipGetDevice(&dev);
                    const unsigned long long bit = 1ull << (dev & 63);
                    if (!(allowed[slot].load(std::memory_order_acquire) & bit)) {
                        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                            (void)hipGetLastError();
                            e = hipErrorNotSupported;
                            return;
                        }
                        allowed[slot].fetch_or(bit, std::memory_order_release);
                    }
                }
                hipLaunchKernelGGL(kernel, dim3((unsigned)nblk), dim3(256), lds, stream, p);
                e = hipGetLastError();
            };
            if (lines == 64) {
                if (p.transpose) go(prefilter_line_tile_kernel<64, true>, 0);
                else go(prefilter_line_tile_kernel<64, false>, 1);
            } else if (lines == 16) {
                if (p.transpose) go(prefilter_line_tile_kernel<16, true>, 2);
                else go(prefilter_line_tile_kernel<16, false>, 3);
            } else {
                if (p.transpose) go(prefilter_line_tile_kernel<4, true>, 4);
                else go(prefilter_line_tile_kernel<4, false>, 5);
            }
            if (e != hipErrorNotSupported)
                return e;
        }
    }
    // what is left with lines of up to 128 samples: one thread per line, working copy in LDS (64 KiB:
    // 256 threads x 32 samples ... 64 x 128), the global scratch buffer is not touched
    if (p.len <= 128 && !no_ldsws) {
        const int blk = p.len <= 32 ? 256 : (p.len <= 64 ? 128 : 64);
        const size_t lds = (size_t)p.len * blk * sizeof(double);
        const int64_t nblk = (p.nlines + blk - 1) / blk;
        if (nblk <= 0x7fffffffLL) {
            const dim3 grid((unsigned)nblk);
            if (p.transpose)
                hipLaunchKernelGGL((prefilter_kernel<true, true>), grid, dim3(blk), lds, stream, p,
                                   (int64_t)0, p.nlines);
            else
                hipLaunchKernelGGL((prefilter_kernel<true, false>), grid, dim3(blk), lds, stream, p, (int64_t)0,
                                   p.nlines);
            return hipGetLastError();
        }
    }
    const int block = 256;
    // p.ws holds p.ws_lines lines; walk the lines in chunks of that size (same stream, so the
    // chunks reuse the scratch buffer in order)
    const int64_t chunk = p.ws_lines > 0 ? p.ws_lines : p.nlines;
    for (int64_t line0 = 0; line0 < p.nlines; line0 += chunk) {
        const int64_t nl = p.nlines - line0 < chunk ? p.nlines - line0 : chunk;
        const dim3 grid((unsigned)((nl + block - 1) / block));
        if (p.transpose)
            hipLaunchKernelGGL((prefilter_kernel<false, true>), grid, dim3(block), 0, stream, p, line0, nl);
        else
            hipLaunchKernelGGL((prefilter_kernel<false, false>), grid, dim3(block), 0, stream, p, line0, nl);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

}  // namespace ed
