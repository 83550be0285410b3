// ed_gridfilter.h -- the whole-grid order-3 prefilter of a small control grid on an LDS copy: ONE definition for
// grid_prefilter_kernel (spline_filter.hip) and for the tables kernel of the tile path (deform_tile.hip), which
// filters the raw grid itself instead of waiting for a launch of its own -- both must produce the same bits.
#pragma once
#include "ed_device.h"
#include "ed_params.h"

namespace ed {

// One line of N samples, element i at c[i * inner], through the recursion of prefilter_kernel -- the very operations of the
// loop version below in the same order (same bits), but on registers: on an LDS copy every step of the chain was a
// read-modify-write round trip (~25 dependent LDS accesses per line and axis: 7 of the geometry kernel's 28 us).
template <int N>
__device__ __forceinline__ void grid_filter_line(double* c, int inner, double z, double gain, double zn1)
{
#pragma clang fp contract(off)
    double v[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        v[i] = c[i * inner] * gain;
    double c0 = v[0] + zn1 * v[N - 1];
    double zi = z;
#pragma unroll
    for (int i = 1; i < N - 1; ++i) {
        c0 += zi * (v[i] + zn1 * v[N - 1 - i]);
        zi *= z;
    }
    c0 /= 1 - zn1 * zn1;
    v[0] = c0;
#pragma unroll
    for (int i = 1; i < N; ++i)
        v[i] += z * v[i - 1];
    v[N - 1] = (z * v[N - 2] + v[N - 1]) * z / (z * z - 1);
#pragma unroll
    for (int i = N - 2; i >= 0; --i)
        v[i] = z * (v[i + 1] - v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i)
        c[i * inner] = v[i];
}

// s[0 .. p.total): the grid as doubles in C order.  Gathers the raw grid (arbitrary strides), filters every grid axis
// but the first in turn with the sequential recursion of prefilter_kernel (one thread per line) and rounds the values
// to the grid's storage dtype after each axis, exactly like the reference's per-axis `output=displacement_f` round
// trip (deform_grid.py:166-169).  All NT threads of the workgroup call it; ends with a barrier.
template <int NT>
__device__ __forceinline__ void grid_prefilter_in_lds(const GridPrefilter& p, double* s, int tid)
{
    // (no fused multiply-adds, whatever the translation unit's default: the recursion rounds like the x86-64 reference
    // build and like spline_filter.hip, which is compiled with -ffp-contract=off)
#pragma clang fp contract(off)
    const int total = p.total;
    for (int e = tid; e < total; e += NT) {
        int r = e;
        int64_t off = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const int q = r / p.shape[d];
            off += (int64_t)(r - q * p.shape[d]) * p.stride_bytes[d];
            r = q;
        }
        s[e] = load_as_double(p.in + off, p.dtype);
    }
    __syncthreads();
    const double z = p.pole, gain = p.gain;
    for (int ax = 1; ax < p.ndim; ++ax) {
        const int n = p.shape[ax];
        int inner = 1;
        for (int d = ax + 1; d < p.ndim; ++d)
            inner *= p.shape[d];
        const int nlines = total / n;
        if (n >= 2) {
            const double zn1 = p.pole_pow[ax];
            for (int line = tid; line < nlines; line += NT) {
                const int outer = line / inner, in = line - outer * inner;
                double* c = s + (int64_t)outer * n * inner + in;     // element i at c[i * inner]
                // (the usual grids -- 2 to 8 control points per axis -- on registers)
                switch (n) {
                case 2: grid_filter_line<2>(c, inner, z, gain, zn1); continue;
                case 3: grid_filter_line<3>(c, inner, z, gain, zn1); continue;
                case 4: grid_filter_line<4>(c, inner, z, gain, zn1); continue;
                case 5: grid_filter_line<5>(c, inner, z, gain, zn1); continue;
                case 6: grid_filter_line<6>(c, inner, z, gain, zn1); continue;
                case 7: grid_filter_line<7>(c, inner, z, gain, zn1); continue;
                case 8: grid_filter_line<8>(c, inner, z, gain, zn1); continue;
                default: break;
                }
                for (int i = 0; i < n; ++i)
                    c[i * inner] *= gain;
                double c0 = c[0] + zn1 * c[(n - 1) * inner];
                double zi = z;
                for (int i = 1; i < n - 1; ++i) {
                    c0 += zi * (c[i * inner] + zn1 * c[(n - 1 - i) * inner]);
                    zi *= z;
                }
                c0 /= 1 - zn1 * zn1;
                c[0] = c0;
                for (int i = 1; i < n; ++i)
                    c[i * inner] += z * c[(i - 1) * inner];
                c[(n - 1) * inner] = (z * c[(n - 2) * inner] + c[(n - 1) * inner]) * z / (z * z - 1);
                for (int i = n - 2; i >= 0; --i)
                    c[i * inner] = z * (c[(i + 1) * inner] - c[i * inner]);
            }
        }
        __syncthreads();
        // round trip through the storage dtype after every axis (plain C cast, as the line
        // buffer write-back does)
        if (p.dtype != EDHIP_F64) {
            for (int e = tid; e < total; e += NT) {
                char tmp[8];
                store_cast(tmp, p.dtype, s[e]);
                s[e] = load_as_double(tmp, p.dtype);
            }
            __syncthreads();
        }
    }
}

}  // namespace ed
