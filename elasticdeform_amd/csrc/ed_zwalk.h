// ed_zwalk.h -- what the z-walk kernels share (deform_k1z.hip: forward; deform_k2z.hip: gradient): the displacement from
// R[o_y][o_x][k_z][c] (contracted over y and x once per call by the geometry kernel) with lane-constant control planes and
// scalar z weights, and the general coordinates of a voxel.
#pragma once

#include <hip/hip_runtime.h>

#include "ed_device.h"
#include "ed_params.h"
#include "ed_tile.h"

namespace ed {
namespace tile {

namespace {

typedef const __attribute__((address_space(4))) int* cint_p;          // constant address space: uniform reads are s_load
typedef const __attribute__((address_space(4))) double* cdbl_p;
typedef const __attribute__((address_space(4))) long long* cll_p;

// What general coordinates, the mirror-mapped staging paths and the fix-up read on top of ZFast: written to the workspace
// by the geometry kernel and read from there on demand (constant address space: scalar loads where they are used) -- as
// a by-value kernel argument its 30 doubles were preloaded into scalar registers and spilled (150 SGPR spills).
struct ZGen {
    int in_len[3], out_len[3], off[3];
    int mode;
    float cval;
    int* hint;                // spill feedback: tiles that do not fit the standard box
    double period[3], inv_period[3];
    double aff[12], offd[3];  // the affine map as the general kernels apply it (crop offset added per voxel)
};
typedef const __attribute__((address_space(4))) ZGen* czgen_p;

// the lane's control planes: 4 taps x 3 components, and the z-table entry they were loaded for
struct ZTaps {
    double r[4][3];
    int key[4];          // byte offsets of the planes held (wave-uniform)
};
// z-table entry of a slice: cubic weights + byte offsets of the control planes in an R column (scalar loads when the
// slice is wave-uniform)
typedef double zv4d __attribute__((ext_vector_type(4)));
typedef int zv4i __attribute__((ext_vector_type(4)));
struct ZEnt {
    zv4d w;
    zv4i idx;
};
__device__ __forceinline__ ZEnt k1z_entry(cdbl_p zt, int oz)
{
    ZEnt e;
    e.w = *reinterpret_cast<const __attribute__((address_space(4))) zv4d*>(zt + (size_t)oz * 6);
    e.idx = *reinterpret_cast<const __attribute__((address_space(4))) zv4i*>(zt + (size_t)oz * 6 + 4);
    return e;
}
// if the walk has entered another control interval: the lane's planes of R
__device__ __forceinline__ void k1z_taps(const ZEnt& e, const char* rcol, ZTaps& tp)
{
    if (e.idx[0] != tp.key[0] || e.idx[1] != tp.key[1] || e.idx[2] != tp.key[2] || e.idx[3] != tp.key[3]) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const double2 a = *reinterpret_cast<const double2*>(rcol + e.idx[l]);
            const double b = *reinterpret_cast<const double*>(rcol + e.idx[l] + 16);
            tp.r[l][0] = a.x;
            tp.r[l][1] = a.y;
            tp.r[l][2] = b;
            tp.key[l] = e.idx[l];
        }
    }
}
__device__ __forceinline__ void k1z_slice(cdbl_p zt, const char* rcol, int oz, ZTaps& tp, double (&zw)[4])
{
    const ZEnt e = k1z_entry(zt, oz);
#pragma unroll
    for (int l = 0; l < 4; ++l)
        zw[l] = e.w[l];
    k1z_taps(e, rcol, tp);
}
__device__ __forceinline__ void k1z_disp(const ZTaps& tp, const double (&zw)[4], double (&d)[3])
{
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        d[h] = zw[0] * tp.r[0][h];
#pragma unroll
        for (int l = 1; l < 4; ++l)
            d[h] = fma(zw[l], tp.r[l][h], d[h]);
    }
}

// general coordinates of one voxel (deform.c:771-824): the arithmetic every tile kernel shares (ed_tile.h)
template <int ORDER, bool AFFINE>
__device__ __forceinline__ bool k1z_coords(czgen_p zn, const double (&d)[3], const int (&b)[3], const double (&P)[3], int* start,
                                           float* frac, int* raw_start = nullptr)
{
    int ci[3];
    bool inr[3];
#pragma unroll
    for (int h = 0; h < 3; ++h)
        inr[h] = coord_axis_fast<ORDER, float>(AFFINE ? P[h] + d[h] : d[h], AFFINE ? 0 : b[h], zn->in_len[h], ci[h], frac[h]);
    if (raw_start) {
#pragma unroll
        for (int h = 0; h < 3; ++h)
            raw_start[h] = ci[h] - ORDER / 2;
    }
    bool cst = false;
    if (!(inr[0] && inr[1] && inr[2])) {
        // one divergent region: the axes along which the source point left the array
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            if (!inr[h])
                cst = coord_axis_mapped<ORDER, float>(AFFINE ? P[h] + d[h] : (double)b[h] + d[h], zn->in_len[h], zn->mode,
                                                      zn->period[h], zn->inv_period[h], ci[h], frac[h]) || cst;
        }
    }
#pragma unroll
    for (int h = 0; h < 3; ++h)
        start[h] = cst ? 0 : ci[h] - ORDER / 2;
    return cst;
}

}  // namespace

}  // namespace tile
}  // namespace ed
