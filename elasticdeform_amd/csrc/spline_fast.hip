// spline_fast.hip -- K3 / K4, fast path: B-spline prefilter (mirror boundary) and its transpose for
// float32 / float64 arrays, spline orders 2 and 3 (one pole; float32 also orders 4 and 5 as a
// cascade of two one-pole passes), lines of at least 64 samples.
//
// Same linear operator as scipy.ndimage.spline_filter1d(mode='mirror') (call sites
// deform_grid.py:160,168,271) and as NI_SplineFilter1DGrad (deform.c:1049-1168), evaluated in a
// form that needs NO scratch memory and parallelises inside a line:
//
//   * with pole z and gain (1 - z)(1 - 1/z) the filter is the symmetric two-sided exponential
//     h[k] = h0 z^|k|, h0 = gain * (-z) / (1 - z^2), applied to the boundary-extended line:
//         s[i] = h0 * (yc[i] + ya[i] - x[i]),   yc[i] = x[i] + z yc[i-1],   ya[i] = x[i] + z ya[i+1]
//     forward : x extended by mirroring (x[-j] = x[j], x[n-1+j] = x[n-1-j])
//     transpose: x extended by zeros, then the two tails folded back:
//                out[i] = s[i] + z^i s[0] + z^(n-1-i) s[n-1]   (0 < i < n-1)
//     (verified against the reference's transposed filter to 2e-15);
//   * |z|^32 < 5e-19 (orders 2, 3), so a recursion started from zero 32 samples early is exact to
//     below fp64 rounding.  Every lane owns one SEGMENT of one line, walks it backwards in blocks of
//     32 outputs, recomputes the causal part of each block from a 32-sample warm-up and carries
//     the anti-causal state across blocks.  All arithmetic fp64, one rounding to the storage dtype
//     per output -- the result agrees with the sequential reference recursion to ~1e-16 relative
//     (not bit-for-bit: the exact kernels in spline_filter.hip remain the bit-comparable path).
//   * lines along a strided axis: adjacent lanes own adjacent lines, so every load / store of a
//     wave is one contiguous row segment.  Lines along the contiguous axis: a lane owns a line and
//     moves its samples as 16-byte vectors.  (These block-recompute kernels serve the lines that
//     are too long for an LDS tile; the default is the whole-line tile kernels further down.)
//   * in place (input == output, how the reference chains the axes, deform_grid.py:158-161): one
//     segment per line; every block only reads samples at or below its own outputs plus the mirror
//     of the line's tail, which is read before anything is written.
//
// Algorithmic bytes: 2 * sizeof(T) per sample per axis; the 32-sample warm-ups are re-reads that
// hit L2.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ed_device.h"
#include "ed_params.h"
#include "ed_workspace.h"

namespace ed {

namespace {

constexpr int kB = 32;       // outputs per block
constexpr int kK = 32;       // warm-up samples
constexpr int kBlock = 256;

struct FastFilter {
    const char* in;
    char* out;
    int64_t len;             // n
    int64_t in_axis_stride, out_axis_stride;      // elements
    int nouter;
    int64_t nlines;
    int64_t outer_len[EDHIP_MAX_DIMS];
    int64_t in_outer_stride[EDHIP_MAX_DIMS];      // elements, last entry = fastest outer axis
    int64_t out_outer_stride[EDHIP_MAX_DIMS];
    int64_t seg_len;         // multiple of kB
    int nseg;                // segments per line
    int transpose;
    double z, h0;
    // device-side window (crop-aware prefilter): 2 ints (w0, w1) per array dimension, or nullptr; the array
    // dimension of the filtered axis and of every outer entry
    const int* win;
    int win_axis;
    int win_outer[EDHIP_MAX_DIMS];
    int dry;                 // 1: every check, no launch (can the whole-line tile kernels take this pass?)
    int in16, out16;         // 16-bit float storage on one side of a float32 pass (1 half, 2 bfloat16; ed_device.h)
};

__device__ __forceinline__ void line_offsets(const FastFilter& p, int64_t line, int64_t& in_off,
                                             int64_t& out_off)
{
    in_off = 0;
    out_off = 0;
    int64_t r = line;
    for (int d = p.nouter - 1; d >= 0; --d) {
        const int64_t q = r / p.outer_len[d];
        const int64_t c = r - q * p.outer_len[d];
        in_off += c * p.in_outer_stride[d];
        out_off += c * p.out_outer_stride[d];
        r = q;
    }
}

// index into the boundary-extended line: >= 0 -> sample index, < 0 -> the sample is zero
__device__ __forceinline__ int64_t ext_index(int64_t j, int64_t n, bool transpose)
{
    if (j >= 0 && j < n)
        return j;
    if (transpose)
        return -1;
    if (j < 0)
        j = -j;
    if (j >= n)
        j = 2 * n - 2 - j;
    return (j >= 0 && j < n) ? j : -1;   // n >= 64 > kK: one reflection is always enough
}

// One block: xs[0 .. kK + kB) are the samples b - kK .. b + kB - 1 (already boundary-extended),
// ya_next is ya[b + kB].  Produces o[0 .. kB) = h0 (yc + ya - x) and the new ya_next = ya[b].
template <typename T>
__device__ __forceinline__ void filter_block(const T (&xs)[kK + kB], double z, double h0,
                                             double& ya_next, double (&o)[kB])
{
    double yc = 0.0;
#pragma unroll
    for (int k = 0; k < kK; ++k)
        yc = (double)xs[k] + z * yc;
#pragma unroll
    for (int k = 0; k < kB; ++k) {
        yc = (double)xs[kK + k] + z * yc;
        o[k] = yc;
    }
    double ya = ya_next;
#pragma unroll
    for (int k = kB - 1; k >= 0; --k) {
        const double x = (double)xs[kK + k];
        ya = x + z * ya;
        o[k] = h0 * (o[k] + ya - x);
    }
    ya_next = ya;
}

// anti-causal warm-up above a segment end e: ya[e] from the kK samples e .. e + kK - 1
template <typename T>
__device__ __forceinline__ double warm_anticausal(const T (&xs)[kK], double z)
{
    double ya = 0.0;
#pragma unroll
    for (int k = kK - 1; k >= 0; --k)
        ya = (double)xs[k] + z * ya;
    return ya;
}

// ---- lines along a strided axis: lane <-> line ---------------------------------------------------
// UNIT: the filtered axis itself is contiguous (element stride 1 in and out).  A lane still owns a
// line, but its 64 / 32 consecutive samples move as 16-byte vector loads / stores.
template <typename T, bool UNIT>
__global__ __launch_bounds__(kBlock) void prefilter_fast_strided_kernel(const FastFilter p)
{
    const int64_t id = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (id >= p.nlines * p.nseg)
        return;
    const int64_t seg = id / p.nlines, line = id - seg * p.nlines;
    int64_t in_off, out_off;
    line_offsets(p, line, in_off, out_off);
    const T* __restrict__ src = reinterpret_cast<const T*>(p.in) + in_off;
    T* dst = reinterpret_cast<T*>(p.out) + out_off;
    const int64_t n = p.len;
    const int64_t a = seg * p.seg_len;
    int64_t e = a + p.seg_len;
    if (e > n)
        e = ((n + kB - 1) / kB) * kB;      // blocks are aligned to multiples of kB; tail is masked
    const double z = p.z, h0 = p.h0;
    const bool tr = p.transpose != 0;

    auto sample = [&](int64_t j) -> T {
        const int64_t i = ext_index(j, n, tr);
        return i >= 0 ? src[UNIT ? i : i * p.in_axis_stride] : (T)0;
    };

    double ya_next;
    {
        T xs[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k)
            xs[k] = sample(e + k);
        ya_next = warm_anticausal(xs, z);
    }
    // transpose: s[n-1] = h0 * yc[n-1] (zero extension above), needed for the right fold
    double s_last = 0.0;
    if (tr && e > n - 1 - kK) {
        double yc = 0.0;
        for (int k = kK - 1; k >= 0; --k)
            yc = (double)sample(n - 1 - k) + z * yc;
        s_last = h0 * yc;
    }
    for (int64_t b = e - kB; b >= a; b -= kB) {
        T xs[kK + kB];
        if (b - kK >= 0 && b + kB <= n) {          // interior block (wave-uniform): plain loads
            const T* q = src + (b - kK) * (UNIT ? 1 : p.in_axis_stride);
#pragma unroll
            for (int k = 0; k < kK + kB; ++k)
                xs[k] = q[UNIT ? k : k * p.in_axis_stride];
        } else {
#pragma unroll
            for (int k = 0; k < kK + kB; ++k)
                xs[k] = sample(b - kK + k);
        }
        double o[kB];
        filter_block(xs, z, h0, ya_next, o);
        if (tr) {
            // fold the tails of the zero-extended result back (see header)
            if (b + kB > n - 1 - kK) {
                double zp = 1.0;       // z^(n-1-i), built upwards from i = n-1
                for (int64_t i = n - 1; i > b + kB - 1; --i)
                    zp *= z;
#pragma unroll
                for (int k = kB - 1; k >= 0; --k) {
                    const int64_t i = b + k;
                    if (i <= n - 1) {
                        if (i > 0 && i < n - 1)
                            o[k] += zp * s_last;
                        zp *= z;
                    }
                }
            }
            if (b == 0) {
                const double s0 = o[0];
                double zp = z;
#pragma unroll
                for (int k = 1; k < kB; ++k) {
                    if (k < n - 1)
                        o[k] += zp * s0;
                    zp *= z;
                }
            }
        }
        if (UNIT && b + kB <= n) {
#pragma unroll
            for (int k = 0; k < kB; ++k)
                dst[b + k] = (T)o[k];
        } else {
#pragma unroll
            for (int k = 0; k < kB; ++k)
                if (b + k < n)
                    dst[(b + k) * (UNIT ? 1 : p.out_axis_stride)] = (T)o[k];
        }
    }
}

// ================================================================================================
// Whole-line tiles (the default whenever a tile of complete lines fits the LDS): a workgroup loads
// C (or R) complete lines into LDS with row-contiguous 16-byte loads -- every sample is read from
// global memory exactly once -- and then every (line, 32-output block) pair is an independent work
// item: causal and anti-causal recursions both restart from a 32-sample warm-up read from LDS, so
// there is no carried state, no sequential dependency between blocks and no global re-read.
// Whole lines per tile also make the kernels safe in place.
//
//   strided axis : tile [32 | n | 32+ halo][C], C = 256 bytes of adjacent lines; lane <-> line,
//                  blocks across waves; the mirror / zero extension is materialised in the halo
//                  rows; outputs go straight from registers to global memory (256-byte rows).
//   contiguous   : tile [R][32 | n | 32+ halo], R in {32, 16, 8} lines; the mirror / zero extension
//                  is materialised in the halo; lane <-> line reads 16-byte vectors (pitch / VEC odd:
//                  conflict-free); outputs return through the tile so that global stores are
//                  row-contiguous 16-byte vectors.
// ================================================================================================
struct LineTile {
    const char* in;
    char* out;
    int n;                       // line length
    int nb;                      // number of 32-output blocks per line
    int64_t in_axis_stride, out_axis_stride;      // elements
    int ncol;                    // strided: extent of the unit-stride outer axis (columns)
    int col_tiles;               // strided: tiles along it
    int nouter;                  // remaining outer axes (strided: without the column axis)
    int64_t outer_len[EDHIP_MAX_DIMS];
    int64_t in_outer_stride[EDHIP_MAX_DIMS];
    int64_t out_outer_stride[EDHIP_MAX_DIMS];
    int64_t nlines;              // contiguous: number of lines
    int rows;                    // contiguous: lines per tile (R)
    int pitch;                   // contiguous: tile row pitch in elements
    int64_t ntiles;
    int transpose;
    int fold;                    // transpose: folded tail terms kept (fold_tails)
    int in16, out16;             // see FastFilter
    unsigned long long* trace;   // profiling only (EDHIP_FILTER_TRACE): phase timestamps of workgroup 0
    double z, h0;
    // device-side window (edhip_spline_filter_axes_window): (w0, w1) per array dimension in device memory; the
    // array dimension of the filtered axis, of the column axis (strided) and of every outer entry.  The host
    // sizes tile and grid for the whole array; the kernel shrinks its geometry to the window when it starts.
    const int* win;
    int win_axis, win_col;
    int win_outer[EDHIP_MAX_DIMS];
};

// The tile kernels' geometry, in registers: the host's (whole array) or -- with a device-side window -- the
// window's: lines outside it are not touched, lines inside are filtered as lines of the window's length (its
// ends are the filter's boundaries; the decay margin around the source box makes that invisible inside the
// box).  Scalars read through readfirstlane and a fixed-size length array with static indices: a mutable copy
// of LineTile went to scratch memory and made every pass 2.5x slower, window or not.
constexpr int kWinOuter = 4;      // outer axes a windowed pass may have (arrays of up to 5 dimensions)
struct TileGeo {
    int n, nb, ncol, col_tiles;
    int64_t ntiles, nlines;
    int64_t in_off, out_off;      // elements: the window's origin
    uint32_t olen[kWinOuter];
};

template <int C>
__device__ __forceinline__ TileGeo tile_geometry(const LineTile& p)
{
    TileGeo g;
    g.n = p.n;
    g.nb = p.nb;
    g.ncol = p.ncol;
    g.col_tiles = p.col_tiles;
    g.ntiles = p.ntiles;
    g.nlines = p.nlines;
    g.in_off = g.out_off = 0;
#pragma unroll
    for (int d = 0; d < kWinOuter; ++d)
        g.olen[d] = d < p.nouter ? (uint32_t)p.outer_len[d] : 1u;
    if (!p.win)
        return g;
    auto w = [&](int dim, int k) { return __builtin_amdgcn_readfirstlane(p.win[2 * dim + k]); };
    const int a0 = w(p.win_axis, 0), a1 = w(p.win_axis, 1);
    g.in_off = (int64_t)a0 * p.in_axis_stride;
    g.out_off = (int64_t)a0 * p.out_axis_stride;
    g.n = a1 - a0;
    g.nb = (g.n + kB - 1) / kB;
    int64_t groups = 1;
#pragma unroll
    for (int d = 0; d < kWinOuter; ++d) {
        if (d < p.nouter) {
            const int o0 = w(p.win_outer[d], 0), o1 = w(p.win_outer[d], 1);
            g.in_off += (int64_t)o0 * p.in_outer_stride[d];
            g.out_off += (int64_t)o0 * p.out_outer_stride[d];
            g.olen[d] = (uint32_t)(o1 - o0);
            groups *= o1 - o0;
        }
    }
    if (C > 0) {      // strided: the column axis has unit stride
        const int c0 = w(p.win_col, 0), c1 = w(p.win_col, 1);
        g.in_off += c0;
        g.out_off += c0;
        g.ncol = c1 - c0;
        g.col_tiles = (g.ncol + C - 1) / C;
        g.ntiles = groups * g.col_tiles;
    } else {
        g.nlines = groups;
        g.ntiles = (groups + p.rows - 1) / p.rows;
    }
    return g;
}

// (32-bit decomposition: a 64-bit integer division costs ~1 us on this machine, and the persistent
// kernels do one or two per tile; the host only takes this path for fewer than 2^31 lines)
__device__ __forceinline__ void tile_offsets(const LineTile& p, const TileGeo& g, int64_t idx, int64_t& in_off,
                                             int64_t& out_off)
{
    in_off = 0;
    out_off = 0;
    uint32_t r = (uint32_t)idx;
    if (p.win) {
        // (windowed pass: at most kWinOuter outer axes, lengths in registers)
#pragma unroll
        for (int d = kWinOuter - 1; d > 0; --d) {
            if (d < p.nouter) {
                const uint32_t len = g.olen[d];
                const uint32_t q = r / len;
                const uint32_t c = r - q * len;
                in_off += (int64_t)c * p.in_outer_stride[d];
                out_off += (int64_t)c * p.out_outer_stride[d];
                r = q;
            }
        }
        if (p.nouter > 0) {
            in_off += (int64_t)r * p.in_outer_stride[0];
            out_off += (int64_t)r * p.out_outer_stride[0];
        }
        return;
    }
    for (int d = p.nouter - 1; d > 0; --d) {
        const uint32_t len = (uint32_t)p.outer_len[d];
        const uint32_t q = r / len;
        const uint32_t c = r - q * len;
        in_off += (int64_t)c * p.in_outer_stride[d];
        out_off += (int64_t)c * p.out_outer_stride[d];
        r = q;
    }
    if (p.nouter > 0) {
        in_off += (int64_t)r * p.in_outer_stride[0];
        out_off += (int64_t)r * p.out_outer_stride[0];
    }
}

// transpose: fold the two tails of the zero-extended result back into block b (see header).
// A = arithmetic type of the tile kernels (see TileArith).  kfold: the folded terms z^m s are dropped from
// m = kfold on (|z|^kfold is below the rounding of A: kK in double, LineTile::fold in float).  When the line
// ends with this block (lengths that are a multiple of the block) s[n-1] is the block's own last output -- above
// n-1 the extension is zero, so the anti-causal sum there is x[n-1] itself and o = h0 (yc + x - x) -- and every
// index is static; other lengths recompute it with s_last_fn.  (The fold used to run in two of a line's eight
// blocks, each with a 32-step recursion of its own, and the waves that got those blocks held the tile back.)
template <typename A, typename SLast>
__device__ __forceinline__ void fold_tails(A (&o)[kB], int b, int n, A z, A h0, int kfold, SLast s_last_fn)
{
    if (n == b + kB) {
        const A s_last = o[kB - 1];
        A zp = z;
#pragma unroll
        for (int m = 1; m < kB; ++m) {
            if (m < kfold && m < n - 1)
                o[kB - 1 - m] += zp * s_last;
            zp *= z;
        }
    } else if (b + kB > n - kfold) {
        const A s_last = h0 * s_last_fn();      // s[n-1] = h0 * sum_k z^k x[n-1-k]
        A zp = 1;              // z^(n-1-i), built upwards from i = n-1
        for (int i = n - 1; i > b + kB - 1; --i)
            zp *= z;
#pragma unroll
        for (int k = kB - 1; k >= 0; --k) {
            const int i = b + k;
            if (i <= n - 1) {
                if (i > 0 && i < n - 1)
                    o[k] += zp * s_last;
                zp *= z;
            }
        }
    }
    if (b == 0) {
        const A s0 = o[0];
        A zp = z;
#pragma unroll
        for (int k = 1; k < kB; ++k) {
            if (k < n - 1 && k < kfold)
                o[k] += zp * s0;
            zp *= z;
        }
    }
}

// one independent block: rd(j) = boundary-extended sample j (as A), j in [b - kK, b + kB + kK)
template <typename A, typename Rd>
__device__ __forceinline__ void block_from_reader(Rd rd, int b, A z, A h0, A (&o)[kB])
{
    // (warm-ups only partly unrolled: full unrolling makes the compiler hoist every LDS read and spill)
    A ya = 0;
#pragma unroll 8
    for (int k = kK - 1; k >= 0; --k)
        ya = fma(z, ya, rd(b + kB + k));
    A yc = 0;
#pragma unroll 8
    for (int k = 0; k < kK; ++k)
        yc = fma(z, yc, rd(b - kK + k));
#pragma unroll
    for (int k = 0; k < kB; ++k) {
        yc = fma(z, yc, rd(b + k));
        o[k] = yc;
    }
#pragma unroll
    for (int k = kB - 1; k >= 0; --k) {
        const A x = rd(b + k);
        ya = fma(z, ya, x);
        o[k] = h0 * (o[k] + ya - x);
    }
}

// Arithmetic of the whole-line tile kernels = the storage type.  float32 volumes are filtered in
// float32: the recursions are contractions (|z| < 0.27 for orders 2 / 3), rounding noise does not
// build up -- measured <= 3e-7 of the line's maximum against the fp64 recursion, an order of
// magnitude inside the 1e-5 float32 budget -- and v_cvt_f64_f32 / v_cvt_f32_f64 run at a quarter of
// the FMA rate, which made the fp64 version of these kernels VALU-bound at two workgroups per CU.
template <typename T>
struct TileArith {
    typedef T type;
};

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
    typedef float __attribute__((ext_vector_type(4))) type;
    static constexpr int N = 4;
};
template <>
struct VecOf<double> {
    typedef double __attribute__((ext_vector_type(2))) type;
    static constexpr int N = 2;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// wait for the next tile's prefetch loads (and this tile's global stores) at every barrier.  The tile
// kernels never communicate through global memory inside a launch, so LDS ordering is all they need.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- strided axis ----------------------------------------------------------------------------------
// Persistent workgroups; with VEC the next tile's rows are already in flight (in registers) while
// the current tile is filtered and stored.
// IN16: the input is stored as 16-bit floats (p.in16) and widened on its way into the tile (VEC only)
template <typename T, int C, bool VEC, bool IN16 = false>
__global__ __launch_bounds__(kBlock, sizeof(T) == 8 ? 1 : 2) void prefilter_tile_strided_kernel(const LineTile p)
{
    static_assert(!IN16 || (VEC && sizeof(T) == 4), "16-bit input: float32 arithmetic, vector loads");
    const TileGeo geo = tile_geometry<C>(p);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem) + kK * C;     // [kK | n32 | kK][C]: sample j at row j
    typedef typename VecOf<T>::type V;
    constexpr int VN = VecOf<T>::N;
    constexpr int CH = C / VN;                         // 16-byte chunks per row
    constexpr int RP = kBlock / CH;                    // rows per pass of the vector loads
    constexpr int NPF = VEC ? (C * (int)sizeof(T) == 256 ? 16 : 18) : 1;   // loads per thread per tile
    const int tid = threadIdx.x;
    const int n = geo.n;
    const int ch = tid % CH, r0 = tid / CH;
    const bool tr = p.transpose != 0;
    typedef typename TileArith<T>::type A;
    const A z = (A)p.z, h0 = (A)p.h0;
    const T* in_base = reinterpret_cast<const T*>(p.in) + (IN16 ? 0 : geo.in_off);
    const unsigned short* in16_base = reinterpret_cast<const unsigned short*>(p.in) + geo.in_off;
    T* out_base = reinterpret_cast<T*>(p.out) + geo.out_off;

    struct Where {
        int64_t in_off, out_off;
        int ncols;
    };
    auto locate = [&](int64_t t) {
        Where w;
        const uint32_t grp = (uint32_t)t / (uint32_t)geo.col_tiles;
        const int col0 = (int)((uint32_t)t - grp * (uint32_t)geo.col_tiles) * C;
        w.ncols = geo.ncol - col0 < C ? geo.ncol - col0 : C;
        tile_offsets(p, geo, grp, w.in_off, w.out_off);
        w.in_off += col0;
        w.out_off += col0;
        return w;
    };
    typedef typename std::conditional<IN16, uint2, V>::type VL;     // what a thread keeps in flight per row
    VL v[NPF];
    // (unconditional loads from clamped addresses: a predicated load sits in its own basic block
    // and the compiler then drains vmcnt before each one, serialising the whole prefetch)
    auto issue = [&](const Where& w) {
        if constexpr (IN16) {
            const unsigned short* src = in16_base + w.in_off + (ch * VN < w.ncols ? ch * VN : 0);
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int r = u * RP + r0;
                v[u] = *reinterpret_cast<const uint2*>(src + (int64_t)(r < n ? r : n - 1) * p.in_axis_stride);
            }
        } else {
            const T* src = in_base + w.in_off + (ch * VN < w.ncols ? ch * VN : 0);   // ncols % VN == 0 (host)
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int r = u * RP + r0;
                v[u] = *reinterpret_cast<const V*>(src + (int64_t)(r < n ? r : n - 1) * p.in_axis_stride);
            }
        }
    };

    int64_t t = blockIdx.x;
    if (t >= geo.ntiles)
        return;
    Where cur = locate(t);
    if (VEC)
        issue(cur);
    for (;;) {
        if (VEC) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int r = u * RP + r0;
                if constexpr (IN16) {
                    V x;
                    x[0] = widen16(v[u].x & 0xffffu, p.in16);
                    x[1] = widen16(v[u].x >> 16, p.in16);
                    x[2] = widen16(v[u].y & 0xffffu, p.in16);
                    x[3] = widen16(v[u].y >> 16, p.in16);
                    if (r < n)
                        *reinterpret_cast<V*>(tile + r * C + ch * VN) = x;
                } else {
                    if (r < n)
                        *reinterpret_cast<V*>(tile + r * C + ch * VN) = v[u];
                }
            }
        } else {
            constexpr int RS = kBlock / C;
            const int c = tid % C, rr = tid / C;
            const T* src = in_base + cur.in_off + c;
            const bool ok = c < cur.ncols;
            for (int rb = 0; rb < n; rb += RS * 8) {
                T x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + u * RS + rr;
                    x[u] = (ok && r < n) ? src[(int64_t)r * p.in_axis_stride] : (T)0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + u * RS + rr;
                    if (r < n)
                        tile[r * C + c] = x[u];
                }
            }
        }
        lds_barrier();
        // halo rows: samples -kK .. -1 and n .. n32 + kK - 1, mirrored (forward) or zero (transpose)
        {
            const int nh = kK + (geo.nb * kB - n) + kK;
            const int cc = tid % C;
            for (int h0_ = tid / C; h0_ < nh; h0_ += 8 * (kBlock / C)) {
                T x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {       // eight independent LDS reads, then the writes
                    const int h = h0_ + u * (kBlock / C);
                    const int j = h < kK ? h - kK : n + (h - kK);
                    const int i = tr ? -1 : (j < 0 ? -j : 2 * n - 2 - j);
                    x[u] = (h < nh && i >= 0) ? tile[i * C + cc] : (T)0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int h = h0_ + u * (kBlock / C);
                    const int j = h < kK ? h - kK : n + (h - kK);
                    if (h < nh)
                        tile[j * C + cc] = x[u];
                }
            }
        }
        const int64_t next = t + gridDim.x;
        const bool has_next = next < geo.ntiles;
        Where nxt = cur;
        if (has_next) {
            nxt = locate(next);
            if (VEC)
                issue(nxt);            // in flight while this tile is filtered and stored
        }
        lds_barrier();

        const int c = tid % C;
        const T* col = tile + c;
        T* dst = out_base + cur.out_off;
        for (int blk = tid / C; blk < geo.nb; blk += kBlock / C) {
            const int b = blk * kB;
            A o[kB];
            block_from_reader([&](int j) { return (A)col[j * C]; }, b, z, h0, o);
            if (tr)
                fold_tails(o, b, n, z, h0, p.fold, [&]() {
                    A yc = 0;
#pragma unroll 8
                    for (int k = kK - 1; k >= 0; --k)
                        yc = fma(z, yc, (A)col[(n - 1 - k) * C]);
                    return yc;
                });
            if (c < cur.ncols) {
                T* q = dst + (int64_t)b * p.out_axis_stride + c;
                if (b + kB <= n) {
#pragma unroll
                    for (int k = 0; k < kB; ++k)
                        q[(int64_t)k * p.out_axis_stride] = (T)o[k];
                } else {
#pragma unroll
                    for (int k = 0; k < kB; ++k)
                        if (b + k < n)
                            q[(int64_t)k * p.out_axis_stride] = (T)o[k];
                }
            }
        }
        if (!has_next)
            break;
        lds_barrier();               // every thread is done reading the tile
        t = next;
        cur = nxt;
    }
}

// ---- contiguous axis -------------------------------------------------------------------------------
// OUT16: the output is stored as 16-bit floats (p.out16), narrowed on its way out of the tile (VEC only)
template <typename T, bool VEC, bool OUT16 = false>
__global__ __launch_bounds__(kBlock, sizeof(T) == 8 ? 1 : 2) void prefilter_tile_contig_kernel(const LineTile p)
{
    static_assert(!OUT16 || (VEC && sizeof(T) == 4), "16-bit output: float32 arithmetic, vector stores");
    const TileGeo geo = tile_geometry<0>(p);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename VecOf<T>::type V;
    typedef typename std::conditional<VEC, V, T>::type W;
    constexpr int VN = VEC ? VecOf<T>::N : 1;
    // chunk loads per thread per tile: R * n <= 8192 samples (host)
    constexpr int NPF = VEC ? 8192 / (kBlock * VN) : 1;
    int64_t* row_off = reinterpret_cast<int64_t*>(smem);          // [parity][in / out][32]
    T* tile = reinterpret_cast<T*>(smem + 1024);                  // [R][pitch], sample j at [kK + j]
    const int tid = threadIdx.x;
    const int n = geo.n, R = p.rows, pitch = p.pitch;
    const int n32 = geo.nb * kB;
    const bool tr = p.transpose != 0;
    typedef typename TileArith<T>::type A;
    const A z = (A)p.z, h0 = (A)p.h0;
    const T* in_base = reinterpret_cast<const T*>(p.in) + geo.in_off;
    T* out_base = reinterpret_cast<T*>(p.out) + geo.out_off;
    const int nc = n / VN;                 // n % VN == 0 (host)
    const int total = R * nc;
    const float inv_nc = 1.0f / (float)nc;

    auto setup_rows = [&](int64_t t, int par) {
        if (tid < R) {
            const int64_t line0 = t * R;
            const int nl = (int)(geo.nlines - line0 < R ? geo.nlines - line0 : R);
            int64_t a, b;
            tile_offsets(p, geo, line0 + (tid < nl ? tid : nl - 1), a, b);
            row_off[par * 64 + tid] = a;
            row_off[par * 64 + 32 + tid] = b;
        }
    };
    W v[NPF];
    auto issue = [&](int par) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            // unconditional, clamped (see the strided kernel)
            const int idx = u * kBlock + tid < total ? u * kBlock + tid : total - 1;
            const int r = (int)(((float)idx + 0.5f) * inv_nc), ch = idx - r * nc;
            v[u] = *reinterpret_cast<const W*>(in_base + row_off[par * 64 + r] + ch * VN);
        }
    };

    int64_t t = blockIdx.x;
    if (t >= geo.ntiles)
        return;
    int tslot = 0;
    auto stamp = [&]() {
        if (p.trace && blockIdx.x == 0 && tid == 0 && tslot < 64)
            p.trace[tslot++] = clock64();
    };
    stamp();
    int par = 0;
    setup_rows(t, 0);
    lds_barrier();
        stamp();
    if (VEC)
        issue(0);
    for (;;) {
        if (VEC) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int idx = u * kBlock + tid;
                if (idx < total) {
                    const int r = (int)(((float)idx + 0.5f) * inv_nc), ch = idx - r * nc;
                    *reinterpret_cast<W*>(tile + r * pitch + kK + ch * VN) = v[u];
                }
            }
        } else {
            for (int base = 0; base < total; base += kBlock * 8) {
                T x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * kBlock + tid;
                    if (idx < total) {
                        const int r = (int)(((float)idx + 0.5f) * inv_nc), ch = idx - r * nc;
                        x[u] = in_base[row_off[par * 64 + r] + ch];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * kBlock + tid;
                    if (idx < total) {
                        const int r = (int)(((float)idx + 0.5f) * inv_nc), ch = idx - r * nc;
                        tile[r * pitch + kK + ch] = x[u];
                    }
                }
            }
        }
        const int64_t next = t + gridDim.x;
        const bool has_next = next < geo.ntiles;
        if (has_next)
            setup_rows(next, par ^ 1);
        lds_barrier();
        stamp();
        // halo: positions -kK .. -1 and n .. n32 + kK - 1, mirrored (forward) or zero (transpose)
        {
            // kBlock / R threads per line, each 8 halo samples per pass: independent reads first
            const int nh = kK + (n32 - n) + kK;
            const int tpr = kBlock / R;
            const int hr = tid / tpr;
            T* hrow = tile + hr * pitch + kK;
            for (int h0_ = tid % tpr; h0_ < nh; h0_ += 8 * tpr) {
                T x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int h = h0_ + u * tpr;
                    const int j = h < kK ? h - kK : n + (h - kK);
                    const int i = tr ? -1 : (j < 0 ? -j : 2 * n - 2 - j);
                    x[u] = (h < nh && i >= 0) ? hrow[i] : (T)0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int h = h0_ + u * tpr;
                    const int j = h < kK ? h - kK : n + (h - kK);
                    if (h < nh)
                        hrow[j] = x[u];
                }
            }
        }
        if (VEC && has_next)
            issue(par ^ 1);            // in flight while this tile is filtered and stored
        lds_barrier();
        stamp();

        // one (line, block) item per thread
        const int r = tid % R, blk = tid / R;
        const bool have = blk < geo.nb;
        A o[kB];
        const int b = blk * kB;
        if (have) {
            const T* row = tile + r * pitch + kK;
            if (VEC) {
                // vector reads, VN samples at a time (pitch / VN is odd: conflict-free across lanes)
                auto rdv = [&](int j) { return *reinterpret_cast<const V*>(row + j); };
                A ya = 0;
#pragma unroll 4
                for (int k = kK - VN; k >= 0; k -= VN) {
                    const V x = rdv(b + kB + k);
#pragma unroll
                    for (int e = VN - 1; e >= 0; --e)
                        ya = fma(z, ya, (A)x[e]);
                }
                A yc = 0;
#pragma unroll 4
                for (int k = 0; k < kK; k += VN) {
                    const V x = rdv(b - kK + k);
#pragma unroll
                    for (int e = 0; e < VN; ++e)
                        yc = fma(z, yc, (A)x[e]);
                }
                T xs[kB];
#pragma unroll
                for (int k = 0; k < kB; k += VN) {
                    const V x = rdv(b + k);
#pragma unroll
                    for (int e = 0; e < VN; ++e) {
                        xs[k + e] = x[e];
                        yc = fma(z, yc, (A)x[e]);
                        o[k + e] = yc;
                    }
                }
#pragma unroll
                for (int k = kB - 1; k >= 0; --k) {
                    const A x = (A)xs[k];
                    ya = fma(z, ya, x);
                    o[k] = h0 * (o[k] + ya - x);
                }
            } else {
                block_from_reader([&](int j) { return (A)row[j]; }, b, z, h0, o);
            }
            if (tr)
                fold_tails(o, b, n, z, h0, p.fold, [&]() {
                    A yc = 0;
                    if (VEC) {      // vector reads here too: scalar ones are 8-way bank conflicted
#pragma unroll 4
                        for (int k = n - kK; k < n; k += VN) {
                            const V x = *reinterpret_cast<const V*>(row + k);
#pragma unroll
                            for (int e = 0; e < VN; ++e)
                                yc = fma(z, yc, (A)x[e]);
                        }
                    } else {
#pragma unroll 8
                        for (int k = kK - 1; k >= 0; --k)
                            yc = fma(z, yc, (A)row[n - 1 - k]);
                    }
                    return yc;
                });
        }
        lds_barrier();
        stamp();          // every thread is done reading the input tile
        if (have) {
            T* row = tile + r * pitch + kK + b;
            if (VEC) {
#pragma unroll
                for (int k = 0; k < kB; k += VN) {
                    V x;
#pragma unroll
                    for (int e = 0; e < VN; ++e)
                        x[e] = (T)o[k + e];
                    *reinterpret_cast<V*>(row + k) = x;
                }
            } else {
#pragma unroll
                for (int k = 0; k < kB; ++k)
                    row[k] = (T)o[k];
            }
        }
        lds_barrier();
        stamp();
        {
            const int64_t line0 = t * R;
            const int nl = (int)(geo.nlines - line0 < R ? geo.nlines - line0 : R);
            constexpr int NST = 8192 / (kBlock * VN);
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int idx = u * kBlock + tid;
                const int ic = idx < total ? idx : total - 1;
                const int rr = (int)(((float)ic + 0.5f) * inv_nc), ch = ic - rr * nc;
                const W x = *reinterpret_cast<const W*>(tile + rr * pitch + kK + ch * VN);
                if constexpr (OUT16) {
                    uint2 pk;
                    pk.x = narrow16(x[0], p.out16) | (narrow16(x[1], p.out16) << 16);
                    pk.y = narrow16(x[2], p.out16) | (narrow16(x[3], p.out16) << 16);
                    if (idx < total && rr < nl)
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + geo.out_off +
                                                  row_off[par * 64 + 32 + rr] + ch * VN) = pk;
                } else {
                    if (idx < total && rr < nl)
                        *reinterpret_cast<W*>(out_base + row_off[par * 64 + 32 + rr] + ch * VN) = x;
                }
            }
        }
        if (!has_next)
            break;
        lds_barrier();
        stamp();          // the tile is free for the next lines
        t = next;
        par ^= 1;
    }
}

constexpr size_t kTileLdsBudget = 80 * 1024;      // two workgroups per CU

// hipFuncSetAttribute is per device (the attribute lives with the device's code object): one
// process may drive several GPUs, so the result is cached per kernel instantiation AND device
template <typename K>
hipError_t allow_large_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024)
        return hipSuccess;
    constexpr int kMaxDev = 64;
    static std::atomic<int> state[kMaxDev];      // 0 unknown, 1 ok, 2 failed (one array per K)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileLdsBudget);
    const int st = state[dev].load(std::memory_order_acquire);
    if (st)
        return st == 1 ? hipSuccess : hipErrorNotSupported;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)kTileLdsBudget);
    state[dev].store(e == hipSuccess ? 1 : 2, std::memory_order_release);
    return e;
}

// returns hipErrorNotSupported when no whole-line tile fits
template <typename T>
hipError_t launch_line_tiles(const FastFilter& f, hipStream_t stream)
{
    constexpr int VN = VecOf<T>::N;
    constexpr int CW = 256 / (int)sizeof(T);       // widest tile: 256-byte rows
    if (f.len > 0x3fffffff || f.nlines > 0x7fffffffLL)
        return hipErrorNotSupported;
    LineTile p;
    memset(&p, 0, sizeof(p));
    p.in = f.in;
    p.out = f.out;
    p.n = (int)f.len;
    p.nb = (p.n + kB - 1) / kB;
    p.in_axis_stride = f.in_axis_stride;
    p.out_axis_stride = f.out_axis_stride;
    p.transpose = f.transpose;
    p.z = f.z;
    p.h0 = f.h0;
    {
        // |z|^fold < 1e-10 in float32 arithmetic (order 3: 18 terms; the first pole of order 5: 28), kK in double
        int fold = kK;
        if (sizeof(T) == 4 && f.z != 0.0) {
            const double az = f.z < 0 ? -f.z : f.z;
            double zp = 1.0;
            for (fold = 0; fold < kK && zp >= 1e-10; ++fold)
                zp *= az;
        }
        p.fold = fold;
    }
    p.win = f.win;
    p.win_axis = f.win_axis;
    p.in16 = f.in16;
    p.out16 = f.out16;
    // persistent grid: two workgroups per CU for float32 (LDS and registers are budgeted for exactly
    // that), one for float64 (its fp64 state does not fit 256 registers next to the prefetch)
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        ncu = 256;
    int wgs_per_cu = sizeof(T) == 8 ? 1 : 2;
    if (const char* w = ed_env("EDHIP_TILE_WGS"))
        wgs_per_cu = atoi(w) > 0 ? atoi(w) : wgs_per_cu;
    const int64_t resident = (int64_t)ncu * wgs_per_cu;
    const bool aligned16 = ((uintptr_t)f.in % 16 == 0) && ((uintptr_t)f.out % 16 == 0);
    [[maybe_unused]] const bool mixed = f.in16 || f.out16;      // served by one kernel each (float32 only), or not at all
    auto strides_vec = [&](int skip) {
        for (int d = 0; d < f.nouter; ++d)
            if (d != skip && (f.in_outer_stride[d] % VN || f.out_outer_stride[d] % VN))
                return false;
        return true;
    };
    const bool contig = f.in_axis_stride == 1 && f.out_axis_stride == 1 && f.nouter > 0;
    if (f.win && f.nouter - (contig ? 0 : 1) > kWinOuter)
        return hipErrorNotSupported;       // (windowed passes keep their outer lengths in registers)
    if (contig) {
        const bool vec = aligned16 && p.n % VN == 0 && strides_vec(-1);
        p.pitch = 2 * kK + p.nb * kB + (vec ? VN : 1);
        int R = 32;
        while (R >= 8 && (R * p.nb > kBlock || 1024 + (size_t)R * p.pitch * sizeof(T) > kTileLdsBudget))
            R /= 2;
        if (R < 8)
            return hipErrorNotSupported;
        p.rows = R;
        p.nlines = f.nlines;
        p.nouter = f.nouter;
        for (int d = 0; d < f.nouter; ++d) {
            p.outer_len[d] = f.outer_len[d];
            p.in_outer_stride[d] = f.in_outer_stride[d];
            p.out_outer_stride[d] = f.out_outer_stride[d];
            p.win_outer[d] = f.win_outer[d];
        }
        const size_t lds = 1024 + (size_t)R * p.pitch * sizeof(T);
        p.ntiles = (f.nlines + R - 1) / R;
        const int64_t nblk = p.ntiles < resident ? p.ntiles : resident;
        if (lds > kTileLdsBudget)
            return hipErrorNotSupported;
        if (mixed && (f.in16 || !vec || sizeof(T) != 4))
            return hipErrorNotSupported;      // (16-bit storage: the strided kernel reads it, this one writes it)
        if (f.dry)
            return hipSuccess;
        if constexpr (sizeof(T) == 4) {
            if (f.out16) {
                const hipError_t attr = allow_large_lds(prefilter_tile_contig_kernel<T, true, true>, kTileLdsBudget);
                if (attr != hipSuccess)
                    return hipErrorNotSupported;
                hipLaunchKernelGGL((prefilter_tile_contig_kernel<T, true, true>), dim3((unsigned)nblk), dim3(kBlock),
                                   lds, stream, p);
                return hipGetLastError();
            }
        }
        if (vec) {
            const hipError_t attr = allow_large_lds(prefilter_tile_contig_kernel<T, true>, kTileLdsBudget);
            if (attr != hipSuccess)
                return hipErrorNotSupported;
            if (ed_env("EDHIP_FILTER_TRACE")) {
                static unsigned long long* buf = nullptr;
                if (!buf)
                    (void)hipMalloc((void**)&buf, 64 * 8);
                (void)hipMemsetAsync(buf, 0, 64 * 8, stream);
                p.trace = buf;
                hipLaunchKernelGGL((prefilter_tile_contig_kernel<T, true>), dim3((unsigned)nblk), dim3(kBlock),
                                   lds, stream, p);
                unsigned long long h[64];
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
                fprintf(stderr, "trace:");
                for (int i = 1; i < 64 && h[i]; ++i)
                    fprintf(stderr, " %llu", h[i] - h[i - 1]);
                fprintf(stderr, "\n");
                return hipGetLastError();
            }
            hipLaunchKernelGGL((prefilter_tile_contig_kernel<T, true>), dim3((unsigned)nblk), dim3(kBlock),
                               lds, stream, p);
        } else {
            const hipError_t attr = allow_large_lds(prefilter_tile_contig_kernel<T, false>, kTileLdsBudget);
            if (attr != hipSuccess)
                return hipErrorNotSupported;
            hipLaunchKernelGGL((prefilter_tile_contig_kernel<T, false>), dim3((unsigned)nblk), dim3(kBlock),
                               lds, stream, p);
        }
        return hipGetLastError();
    }
    // strided: the column axis is an outer axis with unit stride in and out
    int cax = -1;
    for (int d = f.nouter - 1; d >= 0; --d)
        if (f.in_outer_stride[d] == 1 && f.out_outer_stride[d] == 1 && f.outer_len[d] > 1) {
            cax = d;
            break;
        }
    if (cax < 0 || f.outer_len[cax] > 0x3fffffff)
        return hipErrorNotSupported;
    int C = CW;
    const size_t tile_rows = (size_t)(2 * kK + p.nb * kB);
    if (tile_rows * C * sizeof(T) > kTileLdsBudget || f.outer_len[cax] <= C / 2)
        C = CW / 2;
    if (tile_rows * C * sizeof(T) > kTileLdsBudget)
        return hipErrorNotSupported;
    p.ncol = (int)f.outer_len[cax];
    p.col_tiles = (p.ncol + C - 1) / C;
    p.win_col = f.win_outer[cax];
    int64_t groups = 1;
    for (int d = 0; d < f.nouter; ++d) {
        if (d == cax)
            continue;
        p.win_outer[p.nouter] = f.win_outer[d];
        p.outer_len[p.nouter] = f.outer_len[d];
        p.in_outer_stride[p.nouter] = f.in_outer_stride[d];
        p.out_outer_stride[p.nouter] = f.out_outer_stride[d];
        groups *= f.outer_len[d];
        p.nouter++;
    }
    p.ntiles = groups * p.col_tiles;
    const int64_t nblk = p.ntiles < resident ? p.ntiles : resident;
    const bool vec = aligned16 && p.ncol % VN == 0 && strides_vec(cax) &&
                     f.in_axis_stride % VN == 0 && f.out_axis_stride % VN == 0;
    const size_t lds = tile_rows * C * sizeof(T);
    if (mixed && (f.out16 || !vec || C != CW || sizeof(T) != 4))
        return hipErrorNotSupported;
    if (f.dry)
        return hipSuccess;
    if constexpr (sizeof(T) == 4) {
        if (f.in16) {
            const hipError_t attr = allow_large_lds(prefilter_tile_strided_kernel<T, CW, true, true>, kTileLdsBudget);
            if (attr != hipSuccess)
                return hipErrorNotSupported;
            hipLaunchKernelGGL((prefilter_tile_strided_kernel<T, CW, true, true>), dim3((unsigned)nblk), dim3(kBlock),
                               lds, stream, p);
            return hipGetLastError();
        }
    }
#define EDHIP_TILE_STRIDED(CC, VV)                                                                   \
    do {                                                                                             \
        const hipError_t attr =                                                                      \
            allow_large_lds(prefilter_tile_strided_kernel<T, CC, VV>, kTileLdsBudget);              \
        if (attr != hipSuccess)                                                                      \
            return hipErrorNotSupported;                                                             \
        hipLaunchKernelGGL((prefilter_tile_strided_kernel<T, CC, VV>), dim3((unsigned)nblk),         \
                           dim3(kBlock), lds, stream, p);                                            \
    } while (0)
    if (C == CW) {
        if (vec)
            EDHIP_TILE_STRIDED(CW, true);
        else
            EDHIP_TILE_STRIDED(CW, false);
    } else {
        if (vec)
            EDHIP_TILE_STRIDED(CW / 2, true);
        else
            EDHIP_TILE_STRIDED(CW / 2, false);
    }
#undef EDHIP_TILE_STRIDED
    return hipGetLastError();
}

}  // namespace

// host entry: returns hipErrorNotSupported when the case is outside the fast envelope
hipError_t launch_spline_filter_fast(const FilterParams& fp, int order, int ndim, int axis,
                                     const int64_t* shape, const int64_t* in_stride_bytes,
                                     const int64_t* out_stride_bytes, hipStream_t stream, const int* window, bool dry)
{
    if (order < 2 || order > 5)
        return hipErrorNotSupported;
    // a device-side window is served by the whole-line tile kernels only (one pole: orders 2 / 3)
    if (window && (order >= 4 || ed_env("EDHIP_NO_LINE_TILES")))
        return hipErrorNotSupported;
    // 16-bit float storage on ONE side of a float32 pass (orders 2 / 3, whole-line tile kernels only): the first pass
    // of a forward chain widens, the last pass of a chain narrows
    auto kind16 = [](int dt) { return dt == EDHIP_F16 ? 1 : (dt == EDHIP_BF16 ? 2 : 0); };
    const int in16 = fp.out_dtype == EDHIP_F32 ? kind16(fp.in_dtype) : 0;
    const int out16 = fp.in_dtype == EDHIP_F32 ? kind16(fp.out_dtype) : 0;
    const bool mixed = in16 || out16;
    if (mixed && (order >= 4 || ed_env("EDHIP_NO_LINE_TILES")))
        return hipErrorNotSupported;
    if (!mixed && (fp.in_dtype != fp.out_dtype || (fp.in_dtype != EDHIP_F32 && fp.in_dtype != EDHIP_F64)))
        return hipErrorNotSupported;
    if (order >= 4) {
        // Two poles = a cascade of two one-pole filters (mirror-boundary filters commute: both are
        // diagonal in the DCT-I basis; the transposes cascade the same way).  |z_1|^32 = 2e-12
        // (order 5): far below float32 rounding, not below float64's -- float32 only.  The second
        // pass runs in place on the output.
        if (fp.in_dtype != EDHIP_F32 || fp.npoles != 2)
            return hipErrorNotSupported;
        FilterParams a = fp;
        a.npoles = 1;
        a.gain = (1.0 - fp.pole[0]) * (1.0 - 1.0 / fp.pole[0]);
        FilterParams b = fp;
        b.npoles = 1;
        b.pole[0] = fp.pole[1];
        b.gain = (1.0 - fp.pole[1]) * (1.0 - 1.0 / fp.pole[1]);
        // Lines too long for an LDS tile are served by the block-recompute kernels, which can only
        // split a line across lanes when they do not run in place: such arrays go through a dense
        // temporary in the workspace (in -> tmp -> out) instead of (in -> out, out -> out in place).
        int64_t count = 1;
        for (int d = 0; d < ndim; ++d)
            count *= shape[d];
        if (fp.len > 576 && count > 0 && (uint64_t)count * 4 <= ((uint64_t)1 << 30)) {
            hipError_t e = hipSuccess;
            char* ws = (char*)workspace_reserve(stream, kWorkspaceGridBytes + (size_t)count * 4, &e);
            if (ws) {
                int64_t tmp_stride[EDHIP_MAX_DIMS];
                int64_t st = 4;
                for (int d = ndim - 1; d >= 0; --d) {
                    tmp_stride[d] = st;
                    st *= shape[d];
                }
                a.out = ws + kWorkspaceGridBytes;
                b.in = ws + kWorkspaceGridBytes;
                e = launch_spline_filter_fast(a, 3, ndim, axis, shape, in_stride_bytes, tmp_stride, stream);
                if (e != hipSuccess)
                    return e;
                return launch_spline_filter_fast(b, 3, ndim, axis, shape, tmp_stride, out_stride_bytes,
                                                 stream);
            }
            (void)hipGetLastError();
        }
        hipError_t e = launch_spline_filter_fast(a, 3, ndim, axis, shape, in_stride_bytes,
                                                 out_stride_bytes, stream);
        if (e != hipSuccess)
            return e;
        b.in = fp.out;
        return launch_spline_filter_fast(b, 3, ndim, axis, shape, out_stride_bytes, out_stride_bytes,
                                         stream);
    }
    if (fp.len < 64)
        return hipErrorNotSupported;
    const int64_t esz = mixed ? 4 : (fp.in_dtype == EDHIP_F32 ? 4 : 8);
    const int64_t esz_in = in16 ? 2 : esz, esz_out = out16 ? 2 : esz;
    if (((uintptr_t)fp.in % esz_in) || ((uintptr_t)fp.out % esz_out))
        return hipErrorNotSupported;
    for (int d = 0; d < ndim; ++d)
        if (in_stride_bytes[d] % esz_in || out_stride_bytes[d] % esz_out)
            return hipErrorNotSupported;

    FastFilter p;
    memset(&p, 0, sizeof(p));
    p.in = fp.in;
    p.out = fp.out;
    p.len = fp.len;
    p.in_axis_stride = in_stride_bytes[axis] / esz_in;
    p.out_axis_stride = out_stride_bytes[axis] / esz_out;
    p.in16 = in16;
    p.out16 = out16;
    p.nlines = 1;
    for (int d = 0; d < ndim; ++d) {
        if (d == axis)
            continue;
        p.outer_len[p.nouter] = shape[d];
        p.in_outer_stride[p.nouter] = in_stride_bytes[d] / esz_in;
        p.out_outer_stride[p.nouter] = out_stride_bytes[d] / esz_out;
        p.win_outer[p.nouter] = d;
        p.nlines *= shape[d];
        p.nouter++;
    }
    p.win = window;
    p.win_axis = axis;
    p.dry = dry ? 1 : 0;
    if (p.nlines <= 0)
        return hipSuccess;
    p.transpose = fp.transpose;
    // the fast kernels use SciPy's / the reference's pole for the respective direction (they
    // differ in the last ulp, see edhip_api.hip) -- fp.pole[0] already holds the right one
    p.z = fp.pole[0];
    p.h0 = fp.gain * (-p.z) / (1.0 - p.z * p.z);

    // segmentation: in place -> one segment per line; otherwise split lines until there are
    // enough waves to fill the chip (>= ~8 per CU)
    const bool in_place = fp.in == (const char*)fp.out;
    const bool contig = p.in_axis_stride == 1 && p.out_axis_stride == 1 && p.nouter > 0;
    const int64_t nblocks_line = (p.len + kB - 1) / kB;
    int64_t nseg = 1;
    if (!in_place) {
        const int64_t want_threads = (int64_t)256 * 64 * 8;
        while (nseg * 2 <= nblocks_line / 2 && p.nlines * nseg < want_threads)
            nseg *= 2;
    }
    int64_t seg_blocks = (nblocks_line + nseg - 1) / nseg;
    p.seg_len = seg_blocks * kB;
    p.nseg = (int)((nblocks_line + seg_blocks - 1) / seg_blocks);

    if (!ed_env("EDHIP_NO_LINE_TILES")) {
        const hipError_t e = (mixed || fp.in_dtype == EDHIP_F32) ? launch_line_tiles<float>(p, stream)
                                                                 : launch_line_tiles<double>(p, stream);
        if (e != hipErrorNotSupported || window || mixed)
            return e;
        (void)hipGetLastError();
    }
    if (window || mixed)
        return hipErrorNotSupported;
    {
        // block-recompute kernels: lane <-> line (UNIT: the filtered axis itself is contiguous)
        const int64_t threads = p.nlines * p.nseg;
        const int64_t nblk = (threads + kBlock - 1) / kBlock;
        if (nblk > 0x7fffffffLL)
            return hipErrorNotSupported;
        if (contig) {
            if (fp.in_dtype == EDHIP_F32)
                hipLaunchKernelGGL((prefilter_fast_strided_kernel<float, true>), dim3((unsigned)nblk),
                                   dim3(kBlock), 0, stream, p);
            else
                hipLaunchKernelGGL((prefilter_fast_strided_kernel<double, true>), dim3((unsigned)nblk),
                                   dim3(kBlock), 0, stream, p);
        } else {
            if (fp.in_dtype == EDHIP_F32)
                hipLaunchKernelGGL((prefilter_fast_strided_kernel<float, false>), dim3((unsigned)nblk),
                                   dim3(kBlock), 0, stream, p);
            else
                hipLaunchKernelGGL((prefilter_fast_strided_kernel<double, false>), dim3((unsigned)nblk),
                                   dim3(kBlock), 0, stream, p);
        }
    }
    return hipGetLastError();
}

}  // namespace ed
