// ed_exact_coord.h -- the source coordinate of one output voxel in the reference's own evaluation
// order (bit-comparable): used by the exact kernels and by the tie-break path of the label kernel.
#pragma once

#include "ed_device.h"
#include "ed_params.h"

namespace ed {

// displacement at output voxel o: cubic B-spline of the control grid, deform.c:650-758
template <int NAXIS>
__device__ __forceinline__ void eval_displacement(const GridGeom& g, const int64_t* o, double* displ)
{
    // reference arithmetic (x86-64, no FMA): keep the products and sums separate in every TU
#pragma clang fp contract(off)
    double dw[NAXIS][4];
    int64_t dtap[NAXIS][4];   // byte offsets of the 4 taps on each grid axis
#pragma unroll
    for (int k = 0; k < NAXIS; ++k) {
        const double cp = control_coordinate(g.ncp[k], o[k] + g.off[k], g.in_len[k]);
        const int64_t start = window_start(cp, 3);
        const bool edge = start < 0 || start + 3 >= g.ncp[k];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int64_t idx = edge ? mirror_index(start + l, g.ncp[k]) : start + l;
            dtap[k][l] = idx * g.disp_stride[k + 1];
        }
        spline_weights(cp, 3, dw[k]);
    }
    constexpr int kDispTaps = 1 << (2 * NAXIS);
#pragma unroll
    for (int h = 0; h < NAXIS; ++h) {
        const char* base = g.disp + g.disp_stride[0] * h;
        double acc = 0.0;
        for (int t = 0; t < kDispTaps; ++t) {   // lexicographic, last axis fastest (:623-636)
            int64_t offs = 0;
#pragma unroll
            for (int k = 0; k < NAXIS; ++k)
                offs += dtap[k][(t >> (2 * (NAXIS - 1 - k))) & 3];
            double coeff = load_as_double(base + offs, g.disp_dtype);
#pragma unroll
            for (int k = 0; k < NAXIS; ++k)
                coeff *= dw[k][(t >> (2 * (NAXIS - 1 - k))) & 3];
            acc += coeff;
        }
        displ[h] = acc;
    }
}

// The same evaluation with the control grid staged in LDS as doubles, component-major, C order
// (stage_grid_lds): the converted values, the products and the order of the sums are identical to
// eval_displacement's, so the result is bit-equal -- but 64 x naxis dependent global loads per
// voxel become LDS reads (the exact kernels were latency-bound on them).
template <int NAXIS>
__device__ __forceinline__ int stage_grid_lds(const GridGeom& g, double* sgrid)
{
    int cstride[NAXIS];
    int per = 1;
#pragma unroll
    for (int k = NAXIS - 1; k >= 0; --k) {
        cstride[k] = per;
        per *= (int)g.ncp[k];
    }
    for (int e = threadIdx.x; e < per * NAXIS; e += blockDim.x) {
        int r = e % per;
        int64_t offs = g.disp_stride[0] * (e / per);
#pragma unroll
        for (int k = 0; k < NAXIS; ++k) {
            offs += g.disp_stride[k + 1] * (r / cstride[k]);
            r %= cstride[k];
        }
        sgrid[e] = load_as_double(g.disp + offs, g.disp_dtype);
    }
    return per;
}

template <int NAXIS>
__device__ __forceinline__ void eval_displacement_lds(const GridGeom& g, const double* sgrid, int per,
                                                      const int64_t* o, double* displ)
{
    // reference arithmetic (x86-64, no FMA): keep the products and sums separate in every TU
#pragma clang fp contract(off)
    double dw[NAXIS][4];
    int dtap[NAXIS][4];
    int cs = 1;
#pragma unroll
    for (int k = NAXIS - 1; k >= 0; --k) {
        const double cp = control_coordinate(g.ncp[k], o[k] + g.off[k], g.in_len[k]);
        const int64_t start = window_start(cp, 3);
        const bool edge = start < 0 || start + 3 >= g.ncp[k];
#pragma unroll
        for (int l = 0; l < 4; ++l)
            dtap[k][l] = (int)(edge ? mirror_index(start + l, g.ncp[k]) : start + l) * cs;
        spline_weights(cp, 3, dw[k]);
        cs *= (int)g.ncp[k];
    }
    constexpr int kDispTaps = 1 << (2 * NAXIS);
#pragma unroll
    for (int h = 0; h < NAXIS; ++h) {
        double acc = 0.0;
#pragma unroll 16
        for (int t = 0; t < kDispTaps; ++t) {   // lexicographic, last axis fastest (:623-636)
            int offs = h * per;
#pragma unroll
            for (int k = 0; k < NAXIS; ++k)
                offs += dtap[k][(t >> (2 * (NAXIS - 1 - k))) & 3];
            double coeff = sgrid[offs];
#pragma unroll
            for (int k = 0; k < NAXIS; ++k)
                coeff *= dw[k][(t >> (2 * (NAXIS - 1 - k))) & 3];
            acc += coeff;
        }
        displ[h] = acc;
    }
}

// source coordinate before the boundary map, deform.c:771-781
template <int NAXIS>
__device__ __forceinline__ double raw_coordinate(const GridGeom& g, const int64_t* o, int h, double displ)
{
    // reference arithmetic (x86-64, no FMA): keep the products and sums separate in every TU
#pragma clang fp contract(off)
    double cc;
    if (g.has_affine) {
        cc = 0.0;
#pragma unroll
        for (int l = 0; l < NAXIS; ++l)
            cc += g.affine[h * (NAXIS + 1) + l] * (double)o[l];
        cc += g.affine[h * (NAXIS + 1) + NAXIS];
    } else {
        cc = (double)o[h];
    }
    return cc + (double)g.off[h] + displ;
}

}  // namespace ed
