// deform_fast.hip -- K1 (forward gather) and K2 (gradient scatter-add), "fast" arithmetic, for
// float32 / float64 volumes with 1-3 deformed axes and spline orders 0-5.
//
// Same per-voxel pipeline as DeformGrid's hot loop (deform.c:649-1001) -- displacement B-spline,
// optional affine, boundary map, (order+1)^naxis tap gather / scatter -- restructured for CDNA4:
//
//  * coordinates stay fp64 end to end (control-point coordinate, displacement, affine, boundary
//    map, floor, fractional part, basis weights): fp32 coordinates at magnitude 256 already cost
//    3e-5, fp64 VALU is cheap on MI355X;
//  * the displacement spline is evaluated SEPARABLY.  Its weights depend only on the output
//    index along each axis (deform.c:639-647), so for one output row (all deformed indices fixed
//    except the last) the contraction over the slow axes is done once per row by the wave's 64
//    lanes in parallel and parked in LDS:  E[h][j] = sum_{l0,l1} w0[l0] w1[l1] D[h,i0,i1,j].
//    Each voxel then needs only 4 x-taps per component (12 fp64 FMAs in 3-D instead of the
//    reference's 192 multiply-adds).  The slow-axis weights / mirror-mapped control-point
//    indices of the block's rows are staged in LDS by the block prologue (the reference's
//    `dsplvals` table, built per tile instead of per call);
//  * one wave = one output row segment of 64 consecutive voxels along the fastest deformed axis,
//    so output stores and the x-taps of the gathers are coalesced;
//  * taps are accumulated separably (x, then y, then z) in the data's own width: fp32 FMAs for
//    float32 volumes (6e-7 max abs error vs the fp64 reference on white noise, SURVEY.md section
//    7), fp64 for float64 volumes;
//  * K2 uses hardware float atomics (global_atomic_add_f32 / _f64); dX must be zero on entry.
//
// Differences from the reference are rounding only (summation order, fp32 tap arithmetic); the
// parity tests bound them at 1e-5 (float32) / 1e-11 (float64).  Integer and bool volumes never
// come here: they take the exact kernels (deform_exact.hip).
#include <cstdlib>

#include "ed_device.h"
#include "ed_params.h"

namespace ed {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kRows = 32;           // output rows per block (8 per wave)
constexpr int kMaxE = 768;          // naxis * ncp_x doubles of E per wave (LDS: 4 * 6 KiB)

template <typename T>
__device__ __forceinline__ void atomic_add(T* p, T v)
{
    unsafeAtomicAdd(p, v);
}

// element-unit view of IOView for the typed kernels
template <int NAXIS>
struct FastView {
    int64_t in_stride[NAXIS];
    int64_t out_stride[NAXIS];
};

// (four axes: ~190-245 registers, two waves per SIMD; capping the kernel at 168 / 128 registers spills 20-109 of them)
template <typename T, int NAXIS, int ORDER, bool GRAD>
__global__ __launch_bounds__(kBlock) void deform_fast_kernel(const GridGeom g, const IOView v,
                                                             const FastView<NAXIS> fv,
                                                             const int64_t nrows, const int xblocks,
                                                             const int rows)
{
    // rows: output rows per block, a multiple of kWaves up to kRows (fewer for small images, so that
    // the launch still has a few hundred blocks: every row of a wave is a serial chain of grid loads,
    // contraction and gather -- 200x300 with 32 rows per block: 35 blocks, 26 us; with 4: 5 us)
    constexpr int NS = NAXIS - 1;                  // slow (row) axes
    constexpr int NSD = NS > 0 ? NS : 1;
    constexpr int X = NAXIS - 1;                   // fastest deformed axis
    constexpr int NT = ORDER + 1;
    __shared__ double s_w[kRows][NSD][4];          // displacement weights of the slow axes per row
    __shared__ int s_i[kRows][NSD][4];             // mirror-mapped control-point indices
    __shared__ double s_E[kWaves][kMaxE];          // per-wave row contraction of the grid

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xb = blockIdx.x % xblocks;
    const int64_t rb = blockIdx.x / xblocks;
    const int64_t ncpx = g.ncp[X];

    // ---- block prologue: slow-axis displacement tables for this block's rows -----------------
    if (NS > 0) {
        for (int t = tid; t < rows * NS; t += kBlock) {
            const int rr = t / NSD, k = t - rr * NSD;
            int64_t row = rb * rows + rr;
            if (row < nrows) {
                // decompose row -> o_k (last slow axis fastest)
                int64_t ok = 0;
                for (int a = NS - 1; a >= 0; --a) {
                    const int64_t q = row / g.out_len[a];
                    const int64_t c = row - q * g.out_len[a];
                    if (a == k)
                        ok = c;
                    row = q;
                }
                const double cp = control_coordinate(g.ncp[k], ok + g.off[k], g.in_len[k]);
                const int64_t start = window_start(cp, 3);
                const bool edge = start < 0 || start + 3 >= g.ncp[k];
                double w[4];
                spline_weights(cp, 3, w);
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    s_w[rr][k][l] = w[l];
                    s_i[rr][k][l] = (int)(edge ? mirror_index(start + l, g.ncp[k]) : start + l);
                }
            }
        }
    }

    // ---- per-lane displacement taps along x (fixed for the whole block) ----------------------
    const int64_t ox = (int64_t)xb * 64 + lane;
    const bool xvalid = ox < g.out_len[X];
    double wx[4];
    int ix[4];
    {
        const double cp = control_coordinate(ncpx, (xvalid ? ox : 0) + g.off[X], g.in_len[X]);
        const int64_t start = window_start(cp, 3);
        const bool edge = start < 0 || start + 3 >= ncpx;
        spline_weights(cp, 3, wx);
#pragma unroll
        for (int l = 0; l < 4; ++l)
            ix[l] = (int)(edge ? mirror_index(start + l, ncpx) : start + l);
    }
    __syncthreads();

    const T* __restrict__ in = (const T*)v.in;
    T* out = (T*)v.out;
    const int nE = NAXIS * (int)ncpx;

    for (int rr = wave; rr < rows; rr += kWaves) {
        const int64_t row = rb * rows + rr;
        if (row >= nrows)
            break;                                   // wave-uniform
        // row -> slow output indices
        int64_t o[NAXIS];
        {
            int64_t r = row;
#pragma unroll
            for (int a = NS - 1; a >= 0; --a) {
                const int64_t q = r / g.out_len[a];
                o[a] = r - q * g.out_len[a];
                r = q;
            }
            o[X] = ox;
        }

        // ---- E[h][j]: contract the grid over the slow axes, 64 lanes in parallel -------------
        // Four deformed axes with a small control grid: the 64 taps of the three slow axes are the 64 LANES (one
        // independent grid load per lane and element, a butterfly sum per element) instead of 64 dependent loads in
        // each of a dozen busy lanes -- that chain was 95 us per row and 0.38 ms of a 32^4 forward call of ANY order.
        if (NS == 3 && nE <= 64) {
            const int l0 = lane >> 4, l1 = (lane >> 2) & 3, l2 = lane & 3;
            const int64_t offs = g.disp_stride[1] * s_i[rr][0][l0] + g.disp_stride[2] * s_i[rr][NSD > 1 ? 1 : 0][l1] +
                                 g.disp_stride[3] * s_i[rr][NSD > 2 ? 2 : 0][l2];
            const double wprod = s_w[rr][0][l0] * s_w[rr][NSD > 1 ? 1 : 0][l1] * s_w[rr][NSD > 2 ? 2 : 0][l2];
            double mine = 0.0;
            for (int e0 = 0; e0 < nE; e0 += 4) {
                double a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {       // four independent loads in flight, four interleaved butterflies
                    const int e = e0 + u < nE ? e0 + u : nE - 1;
                    const int h = e / (int)ncpx, j = e - h * (int)ncpx;
                    a[u] = load_as_double(g.disp + g.disp_stride[0] * h + g.disp_stride[NAXIS] * j + offs, g.disp_dtype) * wprod;
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        a[u] += __shfl_xor(a[u], m);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    mine = lane == e0 + u ? a[u] : mine;
            }
            if (lane < nE)
                s_E[wave][lane] = mine;
        } else
        for (int e = lane; e < nE; e += 64) {
            const int h = e / (int)ncpx, j = e - h * (int)ncpx;
            const char* base = g.disp + g.disp_stride[0] * h + g.disp_stride[NAXIS] * j;
            double acc = 0.0;
            if (NS == 0) {
                acc = load_as_double(base, g.disp_dtype);
            } else if constexpr (NS == 3) {
                // (large control grids with four deformed axes: 64 taps, four at a time -- unrolled in full they held
                // 128 registers and left the whole kernel one wave per SIMD)
#pragma unroll 4
                for (int t = 0; t < 64; ++t) {
                    int64_t offs = 0;
                    double wprod = 1.0;
#pragma unroll
                    for (int a = 0; a < NS; ++a) {
                        const int l = (t >> (2 * (NS - 1 - a))) & 3;
                        offs += g.disp_stride[a + 1] * s_i[rr][a][l];
                        wprod *= s_w[rr][a][l];
                    }
                    acc += load_as_double(base + offs, g.disp_dtype) * wprod;
                }
            } else {
                constexpr int NTAP = 1 << (2 * NS);
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    int64_t offs = 0;
                    double wprod = 1.0;
#pragma unroll
                    for (int a = 0; a < NS; ++a) {
                        const int l = (t >> (2 * (NS - 1 - a))) & 3;
                        offs += g.disp_stride[a + 1] * s_i[rr][a][l];
                        wprod *= s_w[rr][a][l];
                    }
                    acc += load_as_double(base + offs, g.disp_dtype) * wprod;
                }
            }
            s_E[wave][e] = acc;
        }
        // same-wave LDS write -> read: DS ops of one wave execute in order; only keep the compiler
        // from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (xvalid) {
            // ---- displacement, source coordinate, boundary map (fp64) -------------------------
            double cc[NAXIS];
            bool constant = false;
#pragma unroll
            for (int h = 0; h < NAXIS; ++h) {
                double d = 0.0;
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    d += wx[l] * s_E[wave][h * (int)ncpx + ix[l]];
                double c;
                if (g.has_affine) {
                    c = g.affine[h * (NAXIS + 1) + NAXIS];
#pragma unroll
                    for (int l = 0; l < NAXIS; ++l)
                        c += g.affine[h * (NAXIS + 1) + l] * (double)o[l];
                } else {
                    c = (double)o[h];
                }
                c = map_coordinate(c + (double)g.off[h] + d, g.in_len[h], v.mode);
                constant = constant || !(c > -1.0);
                cc[h] = c;
            }

            // ---- window, tap offsets (elements), weights ------------------------------------
            int64_t tap[NAXIS][NT];
            T w[NAXIS][NT];
            bool xrun = false;       // x taps are consecutive elements in memory
            if (!constant) {
#pragma unroll
                for (int h = 0; h < NAXIS; ++h) {
                    const int64_t start = window_start(cc[h], ORDER);
                    const bool edge = start < 0 || start + ORDER >= g.in_len[h];
#pragma unroll
                    for (int l = 0; l < NT; ++l)
                        tap[h][l] = (edge ? mirror_index(start + l, g.in_len[h]) : start + l) *
                                    fv.in_stride[h];
                    if (h == X)
                        xrun = !edge && fv.in_stride[X] == 1;
                    if (ORDER > 0) {
                        double wd[NT];
                        spline_weights(cc[h], ORDER, wd);
#pragma unroll
                        for (int l = 0; l < NT; ++l)
                            w[h][l] = (T)wd[l];
                    } else {
                        w[h][0] = (T)1;
                    }
                }
            }
            (void)xrun;

            int64_t obase = 0;
#pragma unroll
            for (int k = 0; k < NAXIS; ++k)
                obase += fv.out_stride[k] * o[k];

            // ---- steps: the non-deformed axes reuse coordinates and weights (deform.c:828-838)
            for (int64_t ss = 0; ss < v.nsteps; ++ss) {
                int64_t in_off = 0, out_off = obase;
                {
                    int64_t r = ss;
                    for (int l = 0; l < v.nstep; ++l) {
                        const int64_t q = r / v.step_len[l];
                        const int64_t c = r - q * v.step_len[l];
                        in_off += v.in_step_stride[l] * c;      // element units (host converted)
                        out_off += v.out_step_stride[l] * c;
                        r = q;
                    }
                }
                if (!GRAD) {
                    T val;
                    if (constant) {
                        val = (T)v.cval;
                    } else {
                        const T* p = in + in_off;
                        if constexpr (NAXIS == 1) {
                            T a0 = 0;
#pragma unroll
                            for (int l = 0; l < NT; ++l)
                                a0 += w[0][l] * p[tap[0][l]];
                            val = a0;
                        } else if constexpr (NAXIS == 2) {
                            T a0 = 0;
#pragma unroll
                            for (int l0 = 0; l0 < NT; ++l0) {
                                const T* p0 = p + tap[0][l0];
                                T a1 = 0;
#pragma unroll
                                for (int l1 = 0; l1 < NT; ++l1)
                                    a1 += w[X][l1] * p0[tap[X][l1]];
                                a0 += w[0][l0] * a1;
                            }
                            val = a0;
                        } else if constexpr (NAXIS == 3) {
                            T a0 = 0;
#pragma unroll
                            for (int l0 = 0; l0 < NT; ++l0) {
                                const T* p0 = p + tap[0][l0];
                                T a1 = 0;
#pragma unroll
                                for (int l1 = 0; l1 < NT; ++l1) {
                                    const T* p1 = p0 + tap[1][l1];
                                    T a2 = 0;
#pragma unroll
                                    for (int l2 = 0; l2 < NT; ++l2)
                                        a2 += w[X][l2] * p1[tap[X][l2]];
                                    a1 += w[1][l1] * a2;
                                }
                                a0 += w[0][l0] * a1;
                            }
                            val = a0;
                        } else {
                            // four deformed axes (round 4; the reference takes any number in one loop,
                            // _deform_grid.c:158-175): (order+1)^4 taps.  The two slow axes are ROLLED loops that pick
                            // their tap with a select chain, the (order+1)^2 inner taps are unrolled: fully unrolled
                            // the kernel held 256 registers (two waves per SIMD) and every load's latency showed.
                            T a0 = 0;
#pragma unroll 1
                            for (int l0 = 0; l0 < NT; ++l0) {
                                int64_t t0 = tap[0][0];
                                T w0 = w[0][0];
#pragma unroll
                                for (int q = 1; q < NT; ++q) {
                                    t0 = l0 == q ? tap[0][q] : t0;
                                    w0 = l0 == q ? w[0][q] : w0;
                                }
                                T a1 = 0;
#pragma unroll 1
                                for (int l1 = 0; l1 < NT; ++l1) {
                                    int64_t t1 = tap[1][0];
                                    T w1 = w[1][0];
#pragma unroll
                                    for (int q = 1; q < NT; ++q) {
                                        t1 = l1 == q ? tap[1][q] : t1;
                                        w1 = l1 == q ? w[1][q] : w1;
                                    }
                                    const T* p1 = p + (t0 + t1);
                                    T a2 = 0;
#pragma unroll
                                    for (int l2 = 0; l2 < NT; ++l2) {
                                        const T* p2 = p1 + tap[2][l2];
                                        T a3 = 0;
#pragma unroll
                                        for (int l3 = 0; l3 < NT; ++l3)
                                            a3 += w[X][l3] * p2[tap[X][l3]];
                                        a2 += w[2][l2] * a3;
                                    }
                                    a1 += w1 * a2;
                                }
                                a0 += w0 * a1;
                            }
                            val = a0;
                        }
                    }
                    out[out_off] = val;
                } else if (!constant) {
                    T* p = const_cast<T*>(in) + in_off;
                    const T grad = out[out_off];
                    if constexpr (NAXIS == 1) {
#pragma unroll
                        for (int l = 0; l < NT; ++l)
                            atomic_add(p + tap[0][l], grad * w[0][l]);
                    } else if constexpr (NAXIS == 2) {
#pragma unroll
                        for (int l0 = 0; l0 < NT; ++l0) {
                            const T g0 = grad * w[0][l0];
                            T* p0 = p + tap[0][l0];
#pragma unroll
                            for (int l1 = 0; l1 < NT; ++l1)
                                atomic_add(p0 + tap[X][l1], g0 * w[X][l1]);
                        }
                    } else if constexpr (NAXIS == 3) {
#pragma unroll
                        for (int l0 = 0; l0 < NT; ++l0) {
                            const T g0 = grad * w[0][l0];
                            T* p0 = p + tap[0][l0];
#pragma unroll
                            for (int l1 = 0; l1 < NT; ++l1) {
                                const T g1 = g0 * w[1][l1];
                                T* p1 = p0 + tap[1][l1];
#pragma unroll
                                for (int l2 = 0; l2 < NT; ++l2)
                                    atomic_add(p1 + tap[X][l2], g1 * w[X][l2]);
                            }
                        }
                    } else {
#pragma unroll 1
                        for (int l0 = 0; l0 < NT; ++l0) {
                            int64_t t0 = tap[0][0];
                            T w0 = w[0][0];
#pragma unroll
                            for (int q = 1; q < NT; ++q) {
                                t0 = l0 == q ? tap[0][q] : t0;
                                w0 = l0 == q ? w[0][q] : w0;
                            }
                            const T g0 = grad * w0;
#pragma unroll 1
                            for (int l1 = 0; l1 < NT; ++l1) {
                                int64_t t1 = tap[1][0];
                                T w1 = w[1][0];
#pragma unroll
                                for (int q = 1; q < NT; ++q) {
                                    t1 = l1 == q ? tap[1][q] : t1;
                                    w1 = l1 == q ? w[1][q] : w1;
                                }
                                const T g1 = g0 * w1;
                                T* p1 = p + (t0 + t1);
#pragma unroll
                                for (int l2 = 0; l2 < NT; ++l2) {
                                    const T g2 = g1 * w[2][l2];
                                    T* p2 = p1 + tap[2][l2];
#pragma unroll
                                    for (int l3 = 0; l3 < NT; ++l3)
                                        atomic_add(p2 + tap[X][l3], g2 * w[X][l3]);
                                }
                            }
                        }
                    }
                }
            }
        }
        // the next row of this wave overwrites s_E[wave]: all lanes are past their reads here
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// B-spline basis weights from the fractional offset in the data's width (closed forms of
// deform.c:160-268, last weight = 1 - sum of the others) -- same as the tile kernels use.
template <typename T, int ORDER>
__device__ __forceinline__ void weights_from_frac2(T x, T* w)
{
    const T z = (T)1 - x;
    if (ORDER == 1) {
        w[0] = z;
    } else if (ORDER == 2) {
        w[1] = (T)0.75 - x * x;
        const T y = (T)0.5 - x;
        w[0] = (T)0.5 * y * y;
    } else if (ORDER == 3) {
        w[1] = (x * x * (x - (T)2) * (T)3 + (T)4) * (T)(1.0 / 6.0);
        w[2] = (z * z * (z - (T)2) * (T)3 + (T)4) * (T)(1.0 / 6.0);
        w[0] = z * z * z * (T)(1.0 / 6.0);
    } else if (ORDER == 4) {
        T t = x * x;
        w[2] = t * (t * (T)0.25 - (T)0.625) + (T)(115.0 / 192.0);
        T y = (T)1 + x;
        w[1] = y * (y * (y * ((T)5 - y) * (T)(1.0 / 6.0) - (T)1.25) + (T)(5.0 / 24.0)) + (T)(55.0 / 96.0);
        w[3] = z * (z * (z * ((T)5 - z) * (T)(1.0 / 6.0) - (T)1.25) + (T)(5.0 / 24.0)) + (T)(55.0 / 96.0);
        y = (T)0.5 - x;
        t = y * y;
        w[0] = t * t * (T)(1.0 / 24.0);
    } else {
        T t = x * x;
        w[2] = t * (t * ((T)0.25 - x * (T)(1.0 / 12.0)) - (T)0.5) + (T)0.55;
        t = z * z;
        w[3] = t * (t * ((T)0.25 - z * (T)(1.0 / 12.0)) - (T)0.5) + (T)0.55;
        T y = x + (T)1;
        w[1] = y * (y * (y * (y * (y * (T)(1.0 / 24.0) - (T)0.375) + (T)1.25) - (T)1.75) + (T)0.625) + (T)0.425;
        const T zz = z + (T)1;
        w[4] = zz * (zz * (zz * (zz * (zz * (T)(1.0 / 24.0) - (T)0.375) + (T)1.25) - (T)1.75) + (T)0.625) + (T)0.425;
        y = (T)1 - x;
        t = y * y;
        w[0] = y * t * t * (T)(1.0 / 120.0);
    }
    T last = (T)1;
#pragma unroll
    for (int i = 0; i < ORDER; ++i)
        last -= w[i];
    w[ORDER] = last;
}

// ================================================================================================
// K2 in 2-D with an LDS accumulator (float32 / float64, orders 1-5).  The generic kernel above
// sends every tap to global memory as a float atomic: (order+1)^2 of them per pixel, 268M for a
// 4096^2 image at order 3.  Here a block of kRows2 x 64 output pixels first finds the bounding box
// of all its tap windows (unmapped tap-index space, like the 3-D tile kernel), scatters into a
// fixed-point LDS box with integer atomics (ds_add_u32 / _u64: ~40x the rate of ds_add_f32 on this
// machine) and flushes every touched source element with ONE global atomic, mirror-mapped the way
// deform.c:795-813 maps the taps of a window that sticks out.  The per-block scale
// (2^31 - 2^10) / (w_max * sum |dY|) cannot overflow: |sum into one cell| <= w_max * sum |dY|.
// Blocks whose box does not fit (strong folding, wrap seams) and non-finite gradients fall back to
// direct global atomics.
// ================================================================================================
constexpr int kRows2 = 16;          // output rows per block: 4 per wave, state kept in registers
constexpr int kBoxCap2 = 6144;      // accumulator cells

template <typename T>
struct Fixed;
template <>
struct Fixed<float> {
    typedef int acc_t;
    typedef unsigned int uacc_t;
    static constexpr double kRange = 2147483648.0 - 1024.0;
    __device__ static acc_t round(float x) { return __float2int_rn(x); }
};
template <>
struct Fixed<double> {
    typedef long long acc_t;
    typedef unsigned long long uacc_t;
    static constexpr double kRange = 4611686018427387904.0;       // 2^62
    __device__ static acc_t round(double x) { return __double2ll_rn(x); }
};

template <typename T, int ORDER>
__global__ __launch_bounds__(kBlock) void deform_fast2_grad_kernel(const GridGeom g, const IOView v,
                                                                   const FastView<2> fv,
                                                                   const int64_t nrows, const int xblocks)
{
    constexpr int NT = ORDER + 1;
    constexpr int NP = kRows2 / kWaves;            // pixels per lane
    typedef typename Fixed<T>::acc_t acc_t;
    typedef typename Fixed<T>::uacc_t uacc_t;
    __shared__ double s_w[kRows2][4];              // displacement weights along y per row
    __shared__ int s_i[kRows2][4];
    __shared__ double s_E[kWaves][kMaxE];
    __shared__ int s_red[4];                       // lo_y, lo_x, hi_y, hi_x
    __shared__ T s_sum[kWaves];
    __shared__ __attribute__((aligned(16))) acc_t s_box[kBoxCap2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xb = blockIdx.x % xblocks;
    const int64_t rb = blockIdx.x / xblocks;
    const int64_t ncpx = g.ncp[1];

    for (int t = tid; t < kRows2; t += kBlock) {
        const int64_t row = rb * kRows2 + t;
        if (row < nrows) {
            const double cp = control_coordinate(g.ncp[0], row + g.off[0], g.in_len[0]);
            const int64_t start = window_start(cp, 3);
            const bool edge = start < 0 || start + 3 >= g.ncp[0];
            double w[4];
            spline_weights(cp, 3, w);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                s_w[t][l] = w[l];
                s_i[t][l] = (int)(edge ? mirror_index(start + l, g.ncp[0]) : start + l);
            }
        }
    }
    if (tid < 4)
        s_red[tid] = tid < 2 ? 0x7fffffff : (int)0x80000000;
    const int64_t ox = (int64_t)xb * 64 + lane;
    const bool xvalid = ox < g.out_len[1];
    double wx[4];
    int ix[4];
    {
        const double cp = control_coordinate(ncpx, (xvalid ? ox : 0) + g.off[1], g.in_len[1]);
        const int64_t start = window_start(cp, 3);
        const bool edge = start < 0 || start + 3 >= ncpx;
        spline_weights(cp, 3, wx);
#pragma unroll
        for (int l = 0; l < 4; ++l)
            ix[l] = (int)(edge ? mirror_index(start + l, ncpx) : start + l);
    }
    __syncthreads();

    T* dx = reinterpret_cast<T*>(const_cast<char*>(v.in));
    const T* __restrict__ dy = reinterpret_cast<const T*>(v.out);
    const int nE = 2 * (int)ncpx;

    // ---- phase A: coordinates of this lane's NP pixels (rows wave, wave + 4, ...) --------------
    int st[NP][2];
    T fr[NP][2];
    int64_t obase[NP];
    bool act[NP];
    int lo[2] = {0x7fffffff, 0x7fffffff}, hi[2] = {(int)0x80000000, (int)0x80000000};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int rr = wave + kWaves * i;
        const int64_t row = rb * kRows2 + rr;
        act[i] = false;
        st[i][0] = st[i][1] = 0;
        fr[i][0] = fr[i][1] = 0;
        obase[i] = 0;
        if (row < nrows) {                           // wave-uniform
            for (int e = lane; e < nE; e += 64) {
                const int h = e / (int)ncpx, j = e - h * (int)ncpx;
                const char* base = g.disp + g.disp_stride[0] * h + g.disp_stride[2] * j;
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    acc += load_as_double(base + g.disp_stride[1] * s_i[rr][l], g.disp_dtype) * s_w[rr][l];
                s_E[wave][e] = acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (xvalid) {
                const int64_t o[2] = {row, ox};
                bool constant = false;
                double cc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double d = 0.0;
#pragma unroll
                    for (int l = 0; l < 4; ++l)
                        d += wx[l] * s_E[wave][h * (int)ncpx + ix[l]];
                    double c;
                    if (g.has_affine)
                        c = g.affine[h * 3 + 2] + g.affine[h * 3] * (double)o[0] + g.affine[h * 3 + 1] * (double)o[1];
                    else
                        c = (double)o[h];
                    c = map_coordinate(c + (double)g.off[h] + d, g.in_len[h], v.mode);
                    constant = constant || !(c > -1.0);
                    cc[h] = c;
                }
                if (!constant) {
                    act[i] = true;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const double fl = floor((ORDER & 1) ? cc[h] : cc[h] + 0.5);
                        st[i][h] = (int)fl - ORDER / 2;
                        fr[i][h] = (T)(cc[h] - fl);
                        lo[h] = min(lo[h], st[i][h]);
                        hi[h] = max(hi[h], st[i][h] + ORDER);
                    }
                }
                obase[i] = fv.out_stride[0] * row + fv.out_stride[1] * ox;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- the block's box -----------------------------------------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int l = lo[h], u = hi[h];
        for (int m = 32; m >= 1; m >>= 1) {
            l = min(l, __shfl_xor(l, m));
            u = max(u, __shfl_xor(u, m));
        }
        if (lane == 0) {
            atomicMin(&s_red[h], l);
            atomicMax(&s_red[2 + h], u);
        }
    }
    __syncthreads();
    const int b0[2] = {s_red[0], s_red[1]};
    const bool any = s_red[2] >= s_red[0];
    const int ext0 = any ? s_red[2] - s_red[0] + 1 : 0, ext1 = any ? s_red[3] - s_red[1] + 1 : 0;
    const int pitch = ext1 | 1;                    // odd: rows of neighbouring waves spread over banks
    const bool fits = any && ext0 > 0 && ext1 > 0 && ext1 < 4096 && (int64_t)ext0 * pitch <= kBoxCap2;
    const int nbox = fits ? ext0 * pitch : 0;
    if (!any)
        return;                                    // nothing to scatter (uniform)

    constexpr double kW1 = ORDER == 1 ? 1.0 : ORDER == 2 ? 0.75 : ORDER == 3 ? 2.0 / 3.0
                           : ORDER == 4 ? 115.0 / 192.0 : 0.55;

    for (int64_t ss = 0; ss < v.nsteps; ++ss) {
        int64_t in_off = 0, out_off = 0;
        {
            int64_t r = ss;
            for (int l = 0; l < v.nstep; ++l) {
                const int64_t q = r / v.step_len[l];
                const int64_t c = r - q * v.step_len[l];
                in_off += v.in_step_stride[l] * c;
                out_off += v.out_step_stride[l] * c;
                r = q;
            }
        }
        T* dst = dx + in_off;
        if (ss > 0)
            __syncthreads();                       // previous step's flush is done with the box
        for (int e = tid; e < nbox; e += kBlock)
            s_box[e] = 0;
        T gval[NP];
        T gm = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            gval[i] = act[i] ? dy[out_off + obase[i]] : (T)0;
            const bool finite = fabs((double)gval[i]) <= 1.7976931348623157e308 && gval[i] == gval[i];
            if (!fits || !finite) {
                // direct global atomics: blocks without a box, inf / NaN gradients
                if (gval[i] != (T)0) {
                    T w0[NT], w1[NT];
                    weights_from_frac2<T, ORDER>(fr[i][0], w0);
                    weights_from_frac2<T, ORDER>(fr[i][1], w1);
#pragma unroll 1
                    for (int t = 0; t < NT * NT; ++t) {
                        const int l0 = t / NT, l1 = t % NT;
                        T wa = w0[0], wb = w1[0];
#pragma unroll
                        for (int l = 1; l < NT; ++l) {
                            wa = l0 == l ? w0[l] : wa;
                            wb = l1 == l ? w1[l] : wb;
                        }
                        const int64_t ys = mirror_index(st[i][0] + l0, g.in_len[0]);
                        const int64_t xs = mirror_index(st[i][1] + l1, g.in_len[1]);
                        atomic_add(dst + (ys * fv.in_stride[0] + xs * fv.in_stride[1]), gval[i] * wa * wb);
                    }
                }
                gval[i] = 0;
            }
            gm += fabs(gval[i]);
        }
        for (int m = 32; m >= 1; m >>= 1)
            gm += __shfl_xor(gm, m);
        if (lane == 0)
            s_sum[wave] = gm;
        __syncthreads();                           // box zeroed, sums known
        const T gtot = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        if (!fits || gtot == (T)0)
            continue;                              // (uniform)
        const T scale = (T)(Fixed<T>::kRange / (kW1 * kW1 * 1.001 * (double)gtot));
        const T inv_scale = (T)1 / scale;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (gval[i] == (T)0)
                continue;
            T w0[NT], w1[NT];
            weights_from_frac2<T, ORDER>(fr[i][0], w0);
            weights_from_frac2<T, ORDER>(fr[i][1], w1);
            acc_t* bp = s_box + ((st[i][0] - b0[0]) * pitch + (st[i][1] - b0[1]));
            const T gs = gval[i] * scale;
#pragma unroll
            for (int l0 = 0; l0 < NT; ++l0) {
                const T g0 = gs * w0[l0];
#pragma unroll
                for (int l1 = 0; l1 < NT; ++l1)
                    atomicAdd(reinterpret_cast<uacc_t*>(bp + l0 * pitch + l1),
                              (uacc_t)Fixed<T>::round(g0 * w1[l1]));
            }
        }
        __syncthreads();                           // all contributions are in
        const float inv_pitch = 1.0f / (float)pitch;
        for (int e = tid; e < nbox; e += kBlock) {
            const acc_t a = s_box[e];
            if (a != 0) {
                const int yr = (int)(((float)e + 0.5f) * inv_pitch), xr = e - yr * pitch;
                const int64_t ys = mirror_index(b0[0] + yr, g.in_len[0]);
                const int64_t xs = mirror_index(b0[1] + xr, g.in_len[1]);
                atomic_add(dst + (ys * fv.in_stride[0] + xs * fv.in_stride[1]), (T)a * inv_scale);
            }
        }
    }
}

// ================================================================================================
// Four deformed axes, gradient (round 5; the reference takes any number of axes in one loop,
// _deform_grid.c:158-175, deform.c:953-995): deform_fast2_grad_kernel's scheme with two more axes.  A block owns a
// tile of 4 x 4 x 4 x 4 output voxels (a tile of 2 x 2 x 4 x 16 has a box of 10 x 10 x 12 x 25 cells under a field
// whose gradient is 0.3: every axis of the box grows with the LONGEST edge of the tile; measured 1.8 ms for 32^4, all
// of it the direct fallback), finds the 4-D box of its tap windows, scatters the
// (order + 1)^4 taps of a voxel into fixed-point LDS cells (integer atomics: 256 per voxel at order 3, where the row
// kernel and the exact kernel issue 256 GLOBAL float atomics) and flushes every touched source element with one
// global atomic, mirror-mapped as deform.c:795-813 maps the taps of a window that sticks out.  The per-block scale
// (2^31 - 2^10) / (w_max^4 * sum |dY|) cannot overflow.  Blocks whose box does not fit and non-finite gradients fall
// back to direct global atomics.  Control grids of up to 6 points along x (the lanes of a wave contract the 64
// taps of the three slow axes, one element of E at a time); larger ones stay where they were.
// ================================================================================================
constexpr int kT4A = 4, kT4B = 4, kT4C = 4, kT4X = 4;       // tile: 64 rows x 4 voxels along x
constexpr int kBoxBytes4 = 40 * 1024;
constexpr int kMaxE4 = 24;                                  // 4 * ncp_x doubles of E per row

template <typename T, int ORDER>
__global__ __launch_bounds__(kBlock) void deform_fast4_grad_kernel(const GridGeom g, const IOView v,
                                                                   const FastView<4> fv, const int xblocks,
                                                                   const int tiles_b, const int tiles_c, const int dbg)
{
    constexpr int NT = ORDER + 1;
    constexpr int kRows4 = kT4A * kT4B * kT4C;
    typedef typename Fixed<T>::acc_t acc_t;
    typedef typename Fixed<T>::uacc_t uacc_t;
    constexpr int kCap = kBoxBytes4 / (int)sizeof(acc_t);
    __shared__ double s_w[kRows4][3][4];           // displacement weights of the slow axes per row
    __shared__ int s_i[kRows4][3][4];
    __shared__ double s_E[kRows4][kMaxE4];         // per-row contraction of the grid over the slow axes
    __shared__ int s_red[8];                       // lo[4], hi[4]
    __shared__ T s_sum[kWaves];
    __shared__ __attribute__((aligned(16))) acc_t s_box[kCap];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xb = blockIdx.x % xblocks;
    int tt = blockIdx.x / xblocks;
    const int tc = tt % tiles_c;
    tt /= tiles_c;
    const int tb = tt % tiles_b;
    const int ta = tt / tiles_b;
    const int64_t ncpx = g.ncp[3];
    const int nE = 4 * (int)ncpx;                  // <= kMaxE4 (host)

    auto row_index = [&](int rr, int k) -> int64_t {       // output index of row rr of the tile along slow axis k
        return k == 0 ? (int64_t)ta * kT4A + (rr >> 4) : (k == 1 ? (int64_t)tb * kT4B + ((rr >> 2) & 3) : (int64_t)tc * kT4C + (rr & 3));
    };
    for (int t = tid; t < kRows4 * 3; t += kBlock) {
        const int rr = t / 3, k = t - rr * 3;
        const int64_t ok = min(row_index(rr, k), g.out_len[k] - 1);
        const double cp = control_coordinate(g.ncp[k], ok + g.off[k], g.in_len[k]);
        const int64_t start = window_start(cp, 3);
        const bool edge = start < 0 || start + 3 >= g.ncp[k];
        double w[4];
        spline_weights(cp, 3, w);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            s_w[rr][k][l] = w[l];
            s_i[rr][k][l] = (int)(edge ? mirror_index(start + l, g.ncp[k]) : start + l);
        }
    }
    if (tid < 8)
        s_red[tid] = tid < 4 ? 0x7fffffff : (int)0x80000000;
    __syncthreads();
    // E of the wave's 16 rows: the 64 taps of the three slow axes are the 64 lanes, a butterfly sum per element
    for (int q = 0; q < kRows4 / kWaves; ++q) {
        const int rr = wave * (kRows4 / kWaves) + q;
        const int l0 = lane >> 4, l1 = (lane >> 2) & 3, l2 = lane & 3;
        const int64_t offs = g.disp_stride[1] * s_i[rr][0][l0] + g.disp_stride[2] * s_i[rr][1][l1] + g.disp_stride[3] * s_i[rr][2][l2];
        const double wprod = s_w[rr][0][l0] * s_w[rr][1][l1] * s_w[rr][2][l2];
        double mine = 0.0;
        for (int e0 = 0; e0 < nE; e0 += 4) {
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u < nE ? e0 + u : nE - 1;
                const int h = e / (int)ncpx, j = e - h * (int)ncpx;
                a[u] = load_as_double(g.disp + g.disp_stride[0] * h + g.disp_stride[4] * j + offs, g.disp_dtype) * wprod;
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    a[u] += __shfl_xor(a[u], m);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                mine = lane == e0 + u ? a[u] : mine;
        }
        if (lane < nE)
            s_E[rr][lane] = mine;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- this lane's voxel: row tid / 4 of the tile (its wave made that row's E), x = tid % 4 ---------------------
    const int rr = tid >> 2;
    int64_t o[4];
    o[0] = row_index(rr, 0);
    o[1] = row_index(rr, 1);
    o[2] = row_index(rr, 2);
    o[3] = (int64_t)xb * kT4X + (tid & 3);
    const bool valid = o[0] < g.out_len[0] && o[1] < g.out_len[1] && o[2] < g.out_len[2] && o[3] < g.out_len[3];
    int st[4] = {0, 0, 0, 0};
    T fr[4] = {0, 0, 0, 0};
    bool act = false;
    int64_t obase = 0;
    int lo[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    int hi[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    if (valid) {
        double wx[4];
        int ix[4];
        {
            const double cp = control_coordinate(ncpx, o[3] + g.off[3], g.in_len[3]);
            const int64_t start = window_start(cp, 3);
            const bool edge = start < 0 || start + 3 >= ncpx;
            spline_weights(cp, 3, wx);
#pragma unroll
            for (int l = 0; l < 4; ++l)
                ix[l] = (int)(edge ? mirror_index(start + l, ncpx) : start + l);
        }
        bool constant = false;
        double cc[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            double d = 0.0;
#pragma unroll
            for (int l = 0; l < 4; ++l)
                d += wx[l] * s_E[rr][h * (int)ncpx + ix[l]];
            double c;
            if (g.has_affine) {
                c = g.affine[h * 5 + 4];
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    c += g.affine[h * 5 + l] * (double)o[l];
            } else {
                c = (double)o[h];
            }
            c = map_coordinate(c + (double)g.off[h] + d, g.in_len[h], v.mode);
            constant = constant || !(c > -1.0);
            cc[h] = c;
        }
        if (!constant) {
            act = true;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const double fl = floor((ORDER & 1) ? cc[h] : cc[h] + 0.5);
                st[h] = (int)fl - ORDER / 2;
                fr[h] = (T)(cc[h] - fl);
                lo[h] = st[h];
                hi[h] = st[h] + ORDER;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            obase += fv.out_stride[k] * o[k];
    }
    // ---- the block's box ---------------------------------------------------------------------------------------
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        int l = lo[h], u = hi[h];
        for (int m = 32; m >= 1; m >>= 1) {
            l = min(l, __shfl_xor(l, m));
            u = max(u, __shfl_xor(u, m));
        }
        if (lane == 0) {
            atomicMin(&s_red[h], l);
            atomicMax(&s_red[4 + h], u);
        }
    }
    __syncthreads();
    const bool any = s_red[4] >= s_red[0];
    if (!any)
        return;                                    // nothing to scatter (uniform)
    int b0[4], ext[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        b0[h] = s_red[h];
        ext[h] = s_red[4 + h] - s_red[h] + 1;
    }
    const int pitch = ext[3] | 1;                  // odd: neighbouring rows spread over the banks
    bool fits = true;
#pragma unroll
    for (int h = 0; h < 4; ++h)
        fits = fits && ext[h] > 0 && ext[h] < 1024;
    const int nrows = fits ? ext[0] * ext[1] * ext[2] : 0;
    fits = fits && (int64_t)nrows * pitch <= kCap;
    const int nbox = fits ? nrows * pitch : 0;

    T* dx = reinterpret_cast<T*>(const_cast<char*>(v.in));
    const T* __restrict__ dy = reinterpret_cast<const T*>(v.out);
    constexpr double kW1 = ORDER == 1 ? 1.0 : ORDER == 2 ? 0.75 : 2.0 / 3.0;
    T w[4][NT];
    if (act) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
            weights_from_frac2<T, ORDER>(fr[h], w[h]);
    }
    // taps of the two slow axes picked with select chains inside rolled loops (fully unrolled: 256 registers)
    auto pick = [&](int h, int l) {
        T r = w[h][0];
#pragma unroll
        for (int q = 1; q < NT; ++q)
            r = l == q ? w[h][q] : r;
        return r;
    };

    for (int64_t ss = 0; ss < v.nsteps; ++ss) {
        int64_t in_off = 0, out_off = 0;
        {
            int64_t r = ss;
            for (int l = 0; l < v.nstep; ++l) {
                const int64_t q = r / v.step_len[l];
                const int64_t c = r - q * v.step_len[l];
                in_off += v.in_step_stride[l] * c;
                out_off += v.out_step_stride[l] * c;
                r = q;
            }
        }
        T* dst = dx + in_off;
        if (ss > 0)
            __syncthreads();                       // previous step's flush is done with the box
        for (int e = tid; e < nbox; e += kBlock)
            s_box[e] = 0;
        T gval = act ? dy[out_off + obase] : (T)0;
        const bool finite = fabs((double)gval) <= 1.7976931348623157e308 && gval == gval;
        if ((!fits || !finite) && gval != (T)0) {
            // direct global atomics: blocks without a box, inf / NaN gradients
#pragma unroll 1
            for (int t = 0; t < NT * NT * NT * NT; ++t) {
                const int l3 = t % NT, l2 = (t / NT) % NT, l1 = (t / (NT * NT)) % NT, l0 = t / (NT * NT * NT);
                const int64_t a0 = mirror_index(st[0] + l0, g.in_len[0]), a1 = mirror_index(st[1] + l1, g.in_len[1]);
                const int64_t a2 = mirror_index(st[2] + l2, g.in_len[2]), a3 = mirror_index(st[3] + l3, g.in_len[3]);
                atomic_add(dst + (a0 * fv.in_stride[0] + a1 * fv.in_stride[1] + a2 * fv.in_stride[2] + a3 * fv.in_stride[3]),
                           gval * pick(0, l0) * pick(1, l1) * pick(2, l2) * pick(3, l3));
            }
            gval = 0;
        }
        if (!fits || !finite)
            gval = 0;
        T gm = fabs(gval);
        for (int m = 32; m >= 1; m >>= 1)
            gm += __shfl_xor(gm, m);
        if (lane == 0)
            s_sum[wave] = gm;
        __syncthreads();                           // box zeroed, sums known
        const T gtot = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        if (!fits || gtot == (T)0)
            continue;                              // (uniform)
        const T scale = (T)(Fixed<T>::kRange / (kW1 * kW1 * kW1 * kW1 * 1.001 * (double)gtot));
        const T inv_scale = (T)1 / scale;
        if (gval != (T)0 && !ED_DBG(dbg, 1)) {       // (ablation 1: no scatter)
            acc_t* bp = s_box + ((((st[0] - b0[0]) * ext[1] + (st[1] - b0[1])) * ext[2] + (st[2] - b0[2])) * pitch + (st[3] - b0[3]));
            const T gs = gval * scale;
#pragma unroll 1
            for (int l0 = 0; l0 < NT; ++l0) {
                const T g0 = gs * pick(0, l0);
#pragma unroll 1
                for (int l1 = 0; l1 < NT; ++l1) {
                    const T g1 = g0 * pick(1, l1);
                    acc_t* p1 = bp + (l0 * ext[1] + l1) * ext[2] * pitch;
#pragma unroll
                    for (int l2 = 0; l2 < NT; ++l2) {
                        const T g2 = g1 * w[2][l2];
#pragma unroll
                        for (int l3 = 0; l3 < NT; ++l3)
                            atomicAdd(reinterpret_cast<uacc_t*>(p1 + l2 * pitch + l3), (uacc_t)Fixed<T>::round(g2 * w[3][l3]));
                    }
                }
            }
        }
        __syncthreads();                           // all contributions are in
        // flush: 16 lanes per box row
        {
            const int grp = tid >> 4, sub = tid & 15;
            const float inv2 = 1.0f / (float)ext[2], inv1 = 1.0f / (float)ext[1];
            for (int row = grp; row < nrows; row += kBlock / 16) {
                const int t1 = (int)(((float)row + 0.5f) * inv2), r2 = row - t1 * ext[2];
                const int r0 = (int)(((float)t1 + 0.5f) * inv1), r1 = t1 - r0 * ext[1];
                const int64_t base = mirror_index(b0[0] + r0, g.in_len[0]) * fv.in_stride[0] +
                                     mirror_index(b0[1] + r1, g.in_len[1]) * fv.in_stride[1] +
                                     mirror_index(b0[2] + r2, g.in_len[2]) * fv.in_stride[2];
                for (int xr = sub; xr < ext[3]; xr += 16) {
                    const acc_t a = s_box[row * pitch + xr];
                    if (ED_DBG(dbg, 2) ? a == (acc_t)0x7ffffff1 : a != 0)       // (ablation 2: no flush atomics)
                        atomic_add(dst + (base + mirror_index(b0[3] + xr, g.in_len[3]) * fv.in_stride[3]), (T)a * inv_scale);
                }
            }
        }
    }
}

template <typename T, int NAXIS, int ORDER>
hipError_t launch_typed(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream)
{
    constexpr int X = NAXIS - 1;
    FastView<NAXIS> fv;
    IOView ve = v;     // step strides converted to element units
    for (int k = 0; k < NAXIS; ++k) {
        fv.in_stride[k] = v.in_stride[k] / (int64_t)sizeof(T);
        fv.out_stride[k] = v.out_stride[k] / (int64_t)sizeof(T);
    }
    for (int l = 0; l < v.nstep; ++l) {
        ve.in_step_stride[l] = v.in_step_stride[l] / (int64_t)sizeof(T);
        ve.out_step_stride[l] = v.out_step_stride[l] / (int64_t)sizeof(T);
    }
    int64_t nrows = 1;
    for (int k = 0; k < NAXIS - 1; ++k)
        nrows *= g.out_len[k];
    const int64_t xblocks = (g.out_len[X] + 63) / 64;
    int rows = kRows;
    while (rows > kWaves && xblocks * ((nrows + rows - 1) / rows) < 1024)
        rows >>= 1;
    const int64_t rblocks = (nrows + rows - 1) / rows;
    const int64_t nblk = xblocks * rblocks;
    if (nblk <= 0)
        return hipSuccess;
    if (nblk > 0x7fffffffLL || xblocks > 0x7fffffffLL)
        return hipErrorInvalidValue;
    if constexpr (NAXIS == 2 && ORDER >= 1) {
        if (gradient && !ed_env("EDHIP_2D_DIRECT_GRAD")) {
            const int64_t rblocks2 = (nrows + kRows2 - 1) / kRows2;
            const int64_t nblk2 = xblocks * rblocks2;
            if (nblk2 > 0x7fffffffLL)
                return hipErrorInvalidValue;
            hipLaunchKernelGGL((deform_fast2_grad_kernel<T, ORDER>), dim3((unsigned)nblk2), dim3(kBlock), 0,
                               stream, g, ve, fv, nrows, (int)xblocks);
            return hipGetLastError();
        }
    }
    // (order 1: 16 taps per voxel -- the row kernel's 16 global atomics are faster, 0.32 against 0.72 ms for 32^4)
    if constexpr (NAXIS == 4 && ORDER >= 2 && ORDER <= 3) {
        if (gradient && 4 * g.ncp[3] <= kMaxE4 && !ed_env("EDHIP_4D_DIRECT_GRAD")) {
            const int64_t ta = (g.out_len[0] + kT4A - 1) / kT4A, tb = (g.out_len[1] + kT4B - 1) / kT4B,
                          tc = (g.out_len[2] + kT4C - 1) / kT4C, xb4 = (g.out_len[3] + kT4X - 1) / kT4X;
            const int64_t nblk4 = ta * tb * tc * xb4;
            if (nblk4 > 0x7fffffffLL || tb > 0x7fffffffLL || tc > 0x7fffffffLL)
                return hipErrorInvalidValue;
            const int dbg4 = ed_env("EDHIP_4D_DBG") ? atoi(ed_env("EDHIP_4D_DBG")) : 0;
            hipLaunchKernelGGL((deform_fast4_grad_kernel<T, ORDER>), dim3((unsigned)nblk4), dim3(kBlock), 0, stream, g, ve,
                               fv, (int)xb4, (int)tb, (int)tc, dbg4);
            return hipGetLastError();
        }
    }
    if (gradient)
        hipLaunchKernelGGL((deform_fast_kernel<T, NAXIS, ORDER, true>), dim3((unsigned)nblk),
                           dim3(kBlock), 0, stream, g, ve, fv, nrows, (int)xblocks, rows);
    else
        hipLaunchKernelGGL((deform_fast_kernel<T, NAXIS, ORDER, false>), dim3((unsigned)nblk),
                           dim3(kBlock), 0, stream, g, ve, fv, nrows, (int)xblocks, rows);
    return hipGetLastError();
}

template <typename T, int NAXIS>
hipError_t launch_order(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream)
{
    switch (v.order) {
    case 0: return launch_typed<T, NAXIS, 0>(g, v, gradient, stream);
    case 1: return launch_typed<T, NAXIS, 1>(g, v, gradient, stream);
    case 2: return launch_typed<T, NAXIS, 2>(g, v, gradient, stream);
    case 3: return launch_typed<T, NAXIS, 3>(g, v, gradient, stream);
    case 4: return launch_typed<T, NAXIS, 4>(g, v, gradient, stream);
    case 5: return launch_typed<T, NAXIS, 5>(g, v, gradient, stream);
    default: return hipErrorInvalidValue;
    }
}

template <typename T>
hipError_t launch_axes(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream)
{
    switch (g.naxis) {
    case 1: return launch_order<T, 1>(g, v, gradient, stream);
    case 2: return launch_order<T, 2>(g, v, gradient, stream);
    case 3: return launch_order<T, 3>(g, v, gradient, stream);
    case 4:
        // (orders 0-3: 256 taps per voxel; orders 4 / 5 -- 625 / 1296 -- stay on the exact kernel)
        switch (v.order) {
        case 0: return launch_typed<T, 4, 0>(g, v, gradient, stream);
        case 1: return launch_typed<T, 4, 1>(g, v, gradient, stream);
        case 2: return launch_typed<T, 4, 2>(g, v, gradient, stream);
        case 3: return launch_typed<T, 4, 3>(g, v, gradient, stream);
        default: return hipErrorNotSupported;
        }
    default: return hipErrorNotSupported;
    }
}

}  // namespace

bool deform_fast_supported(const GridGeom& g, const IOView& v, int gradient)
{
    if (g.naxis < 1 || g.naxis > 4 || (g.naxis == 4 && v.order > 3))
        return false;
    // (four axes, order 3, gradient: the LDS-accumulating kernel of round 5 for control grids of up to 6 points along x;
    // beyond that 256 global atomics per voxel on either kernel -- 32^4 float32: 8.0 ms here against 7.1 ms on the exact
    // kernel, which stays in charge)
    if (g.naxis == 4 && gradient && v.order == 3 && 4 * g.ncp[3] > 24)
        return false;
    if (v.in_dtype != v.out_dtype)
        return false;
    if (v.in_dtype != EDHIP_F32 && v.in_dtype != EDHIP_F64)
        return false;
    const int64_t esz = v.in_dtype == EDHIP_F32 ? 4 : 8;
    if (((uintptr_t)v.in % esz) || ((uintptr_t)v.out % esz))
        return false;
    for (int k = 0; k < g.naxis; ++k)
        if (v.in_stride[k] % esz || v.out_stride[k] % esz)
            return false;
    for (int l = 0; l < v.nstep; ++l)
        if (v.in_step_stride[l] % esz || v.out_step_stride[l] % esz)
            return false;
    if ((int64_t)g.naxis * g.ncp[g.naxis - 1] > kMaxE)
        return false;
    for (int k = 0; k < g.naxis; ++k)
        if (g.ncp[k] > 0x7fffffffLL / 4)
            return false;
    return true;
}

hipError_t launch_deform_fast(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream)
{
    if (!deform_fast_supported(g, v, gradient))
        return hipErrorNotSupported;
    if (v.in_dtype == EDHIP_F32)
        return launch_axes<float>(g, v, gradient, stream);
    return launch_axes<double>(g, v, gradient, stream);
}

}  // namespace ed
