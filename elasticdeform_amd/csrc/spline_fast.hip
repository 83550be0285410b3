// spline_fast.hip -- K3 / K4, fast path: B-spline prefilter (mirror boundary) and its transpose for
// float32 / float64 arrays, spline orders 2 and 3 (one pole), lines of at least 64 samples.
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
//     wave is one contiguous row segment.  Lines along the contiguous axis: a wave owns 64 lines
//     and moves 64 x 64 tiles through LDS (padded pitch) so that global accesses stay row-contiguous.
//   * in place (input == output, how the reference chains the axes, deform_grid.py:158-161): one
//     segment per line; every block only reads samples at or below its own outputs plus the mirror
//     of the line's tail, which is read before anything is written.
//
// Algorithmic bytes: 2 * sizeof(T) per sample per axis; the 32-sample warm-ups are re-reads that
// hit L2.
#include <cstdlib>
#include <cstring>

#include "ed_device.h"
#include "ed_params.h"

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

// ---- lines along the contiguous axis: a wave owns 64 lines, tiles go through LDS --------------------
template <typename T>
constexpr int contig_waves()
{
    return sizeof(T) == 4 ? 2 : 1;       // 64 x 65 x sizeof(T) bytes of LDS per wave, <= 33 KiB per block
}

template <typename T>
__global__ __launch_bounds__(64 * contig_waves<T>()) void prefilter_fast_contig_kernel(
    const FastFilter p)
{
    constexpr int kPitch = kK + kB + 1;                 // 65: conflict-free column reads
    constexpr int kWaves = contig_waves<T>();
    __shared__ T tile[kWaves][64 * kPitch];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* tl = tile[wave];
    const int64_t wid = (int64_t)blockIdx.x * kWaves + wave;
    const int64_t groups = (p.nlines + 63) / 64;
    if (wid >= groups * p.nseg)
        return;
    const int64_t seg = wid / groups, line0 = (wid - seg * groups) * 64;
    const int64_t n = p.len;
    const int64_t a = seg * p.seg_len;
    int64_t e = a + p.seg_len;
    if (e > n)
        e = ((n + kB - 1) / kB) * kB;
    const double z = p.z, h0 = p.h0;
    const bool tr = p.transpose != 0;
    const int nl = (int)((p.nlines - line0) < 64 ? (p.nlines - line0) : 64);   // lines in this group

    // lane r holds the base offsets of line r of the group; the row loops fetch them with readlane
    int64_t my_in, my_out;
    line_offsets(p, line0 + (lane < nl ? lane : nl - 1), my_in, my_out);
    const T* in_base = reinterpret_cast<const T*>(p.in);
    T* out_base = reinterpret_cast<T*>(p.out);
    auto row_offset = [&](int64_t v, int r) -> int64_t {
        const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffff), r);
        const int hi = __builtin_amdgcn_readlane((int)(v >> 32), r);
        return ((int64_t)hi << 32) | (uint32_t)lo;
    };
    auto load_tile = [&](int64_t j0, int count) {
        // rows r = 0..nl-1, columns j0 .. j0 + count - 1 (count <= 64): lane <-> column
        const int64_t i = lane < count ? ext_index(j0 + lane, n, tr) : -1;
        const int64_t col = i * p.in_axis_stride;
        // all row loads of a batch are issued before the first LDS write (fixed trip counts so
        // that the loops unroll; rows beyond nl re-read the last line, harmlessly)
#pragma unroll 1
        for (int r0 = 0; r0 < 64; r0 += 16) {
            T v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t off = row_offset(my_in, r0 + r);
                v[r] = i >= 0 ? in_base[off + col] : (T)0;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tl[(r0 + r) * kPitch + lane] = v[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    double ya_next;
    {
        load_tile(e, kK);
        T xs[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k)
            xs[k] = tl[lane * kPitch + k];
        ya_next = warm_anticausal(xs, z);
        wave_sync();
    }
    double s_last = 0.0;
    if (tr && e > n - 1 - kK) {
        load_tile(n - kK, kK);
        double yc = 0.0;
#pragma unroll
        for (int k = 0; k < kK; ++k)
            yc = (double)tl[lane * kPitch + k] + z * yc;
        s_last = h0 * yc;
        wave_sync();
    }
    for (int64_t b = e - kB; b >= a; b -= kB) {
        load_tile(b - kK, kK + kB);
        T xs[kK + kB];
#pragma unroll
        for (int k = 0; k < kK + kB; ++k)
            xs[k] = tl[lane * kPitch + k];
        double o[kB];
        filter_block(xs, z, h0, ya_next, o);
        if (tr) {
            if (b + kB > n - 1 - kK) {
                double zp = 1.0;
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
        wave_sync();          // every lane is done reading its row of the input tile
#pragma unroll
        for (int k = 0; k < kB; ++k)
            tl[lane * kPitch + k] = (T)o[k];
        wave_sync();
        // store rows: two lines per instruction, 32 contiguous outputs each
        const int half = lane >> 5, col = lane & 31;
#pragma unroll 8
        for (int r0 = 0; r0 < 64; r0 += 2) {
            // rows r0 (lanes 0-31) and r0 + 1 (lanes 32-63)
            const int64_t o0 = row_offset(my_out, r0);
            const int64_t o1 = row_offset(my_out, r0 + 1);
            const int r = r0 + half;
            if (r < nl && b + col < n)
                out_base[(half ? o1 : o0) + (b + col) * p.out_axis_stride] = tl[r * kPitch + col];
        }
        wave_sync();
    }
}

}  // namespace

// host entry: returns hipErrorNotSupported when the case is outside the fast envelope
hipError_t launch_spline_filter_fast(const FilterParams& fp, int order, int ndim, int axis,
                                     const int64_t* shape, const int64_t* in_stride_bytes,
                                     const int64_t* out_stride_bytes, hipStream_t stream)
{
    if (order != 2 && order != 3)
        return hipErrorNotSupported;
    if (fp.in_dtype != fp.out_dtype || (fp.in_dtype != EDHIP_F32 && fp.in_dtype != EDHIP_F64))
        return hipErrorNotSupported;
    if (fp.len < 64)
        return hipErrorNotSupported;
    const int64_t esz = fp.in_dtype == EDHIP_F32 ? 4 : 8;
    if (((uintptr_t)fp.in % esz) || ((uintptr_t)fp.out % esz))
        return hipErrorNotSupported;
    for (int d = 0; d < ndim; ++d)
        if (in_stride_bytes[d] % esz || out_stride_bytes[d] % esz)
            return hipErrorNotSupported;

    FastFilter p;
    memset(&p, 0, sizeof(p));
    p.in = fp.in;
    p.out = fp.out;
    p.len = fp.len;
    p.in_axis_stride = in_stride_bytes[axis] / esz;
    p.out_axis_stride = out_stride_bytes[axis] / esz;
    p.nlines = 1;
    for (int d = 0; d < ndim; ++d) {
        if (d == axis)
            continue;
        p.outer_len[p.nouter] = shape[d];
        p.in_outer_stride[p.nouter] = in_stride_bytes[d] / esz;
        p.out_outer_stride[p.nouter] = out_stride_bytes[d] / esz;
        p.nlines *= shape[d];
        p.nouter++;
    }
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

    if (contig && !getenv("EDHIP_CONTIG_LDS")) {
        const int64_t threads = p.nlines * p.nseg;
        const int64_t nblk = (threads + kBlock - 1) / kBlock;
        if (nblk > 0x7fffffffLL)
            return hipErrorNotSupported;
        if (fp.in_dtype == EDHIP_F32)
            hipLaunchKernelGGL((prefilter_fast_strided_kernel<float, true>), dim3((unsigned)nblk),
                               dim3(kBlock), 0, stream, p);
        else
            hipLaunchKernelGGL((prefilter_fast_strided_kernel<double, true>), dim3((unsigned)nblk),
                               dim3(kBlock), 0, stream, p);
    } else if (contig) {
        const int64_t groups = (p.nlines + 63) / 64;
        const int64_t waves = groups * p.nseg;
        const int wpb = fp.in_dtype == EDHIP_F32 ? contig_waves<float>() : contig_waves<double>();
        const int64_t nblk = (waves + wpb - 1) / wpb;
        if (nblk > 0x7fffffffLL)
            return hipErrorNotSupported;
        if (fp.in_dtype == EDHIP_F32)
            hipLaunchKernelGGL(prefilter_fast_contig_kernel<float>, dim3((unsigned)nblk),
                               dim3(64 * wpb), 0, stream, p);
        else
            hipLaunchKernelGGL(prefilter_fast_contig_kernel<double>, dim3((unsigned)nblk),
                               dim3(64 * wpb), 0, stream, p);
    } else {
        const int64_t threads = p.nlines * p.nseg;
        const int64_t nblk = (threads + kBlock - 1) / kBlock;
        if (nblk > 0x7fffffffLL)
            return hipErrorNotSupported;
        if (fp.in_dtype == EDHIP_F32)
            hipLaunchKernelGGL((prefilter_fast_strided_kernel<float, false>), dim3((unsigned)nblk),
                               dim3(kBlock), 0, stream, p);
        else
            hipLaunchKernelGGL((prefilter_fast_strided_kernel<double, false>), dim3((unsigned)nblk),
                               dim3(kBlock), 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace ed
