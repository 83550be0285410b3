// ed_hot.h -- what the 4-wave kernels of the float32 benchmark case share: K2 (deform_hot.hip, shipped) and the
// round-4 kernels kept for A/B measurements in the profiling build (experiments/deform_hot_r4.hip): strip decoding,
// the strip prologue (x table, Q rows, uniform parameters -> LDS), general coordinates, the wave-level box reduction.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "ed_device.h"
#include "ed_params.h"
#include "ed_tile.h"

#ifndef ED_K2_U1
#define ED_K2_U1 1
#endif
#ifndef ED_K2_U2
#define ED_K2_U2 1
#endif
#define ED_PRAGMA(x) _Pragma(#x)
#define ED_UNROLL(n) ED_PRAGMA(unroll n)

namespace ed {
namespace tile {

namespace {

constexpr int kGradBoxBytes = 24 * 1024;       // K2: fixed-point cells per tile (4 workgroups per CU)

// Bounding box of a wave's tap windows: min of lo[3], max of hi[3] over the 64 lanes with DPP only --
// four row steps (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then
// row_bcast:15 / row_bcast:31 carry the row results upwards so that lane 63 holds the wave's result;
// that lane folds it into the tile's six LDS slots (ds_min_i32 x3, ds_max_i32 x3).  Hand-written:
// the compiler turns every __builtin_amdgcn_update_dpp step into copy + s_nop + v_mov_dpp + v_min
// (4 instructions instead of 1) and wraps the single-lane atomics in a wave-reduction loop -- together
// ~200 instructions per wave and tile, a quarter of the kernel's VALU work.  The six chains are
// interleaved, which also covers the DPP read-after-write hazard (2 wait states) without s_nop.
// All 64 lanes must be active.  The LDS atomics are issued from inline assembly, which the compiler's
// s_waitcnt bookkeeping does not see: every barrier that publishes the slots is preceded by
// lds_atomics_done() (without it the barrier can be passed while they are still in flight -- found
// by tests/fuzz/fuzz_hot.py as rare garbage voxels in the non-pipelined order-4 / 5 builds).
#define ED_RED6(OP, CTRL)                                  \
    "v_min_i32_dpp %0, %0, %0 " CTRL "\n\t"                \
    "v_min_i32_dpp %1, %1, %1 " CTRL "\n\t"                \
    "v_min_i32_dpp %2, %2, %2 " CTRL "\n\t"                \
    "v_max_i32_dpp %3, %3, %3 " CTRL "\n\t"                \
    "v_max_i32_dpp %4, %4, %4 " CTRL "\n\t"                \
    "v_max_i32_dpp %5, %5, %5 " CTRL "\n\t"
__device__ __forceinline__ void box_reduce_to_lds(int* red, int lane, int (&lo)[3], int (&hi)[3])
{
    asm volatile("s_nop 1\n\t"
                 ED_RED6(, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 ED_RED6(, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 ED_RED6(, "row_half_mirror row_mask:0xf bank_mask:0xf")
                 ED_RED6(, "row_mirror row_mask:0xf bank_mask:0xf")
                 ED_RED6(, "row_bcast:15 row_mask:0xa bank_mask:0xf")
                 ED_RED6(, "row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
    if (lane == 63) {
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)red;
        asm volatile("ds_min_i32 %0, %1\n\t"
                     "ds_min_i32 %0, %2 offset:4\n\t"
                     "ds_min_i32 %0, %3 offset:8\n\t"
                     "ds_max_i32 %0, %4 offset:12\n\t"
                     "ds_max_i32 %0, %5 offset:16\n\t"
                     "ds_max_i32 %0, %6 offset:20"
                     :
                     : "v"(addr), "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2])
                     : "memory");
    }
}
#undef ED_RED6
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the output side of a call with 16-bit float storage (HotGeom::io16; element offsets count 16-bit elements)
__device__ __forceinline__ void store_out(float* img, long long off, float val, int io16)
{
    if (io16)
        __builtin_nontemporal_store((unsigned short)narrow16(val, io16), reinterpret_cast<unsigned short*>(img) + off);
    else
        __builtin_nontemporal_store(val, img + off);
}
__device__ __forceinline__ float load_dy(const float* dy, long long off, int io16)
{
    return io16 ? widen16(reinterpret_cast<const unsigned short*>(dy)[off], io16) : dy[off];
}
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// (int)floor(x + 0.5) in one instruction; __float2int_rn is v_rndne_f32 + v_cvt_i32_f32 (64 more VALU
// instructions per voxel in the scatter).  Ties go up instead of to even: exact halves of a
// fixed-point unit, no bias that matters.
__device__ __forceinline__ int round_half_up_i32(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ void lds_atomics_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every wave
// would sit out the round trip of the global stores / atomics it has just issued (about 2-4 us per
// tile) although nothing in the workgroup reads those addresses back.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- strip prologue: x table, Q rows, uniform parameters -> LDS ----------------------------------------
struct HotStrip {
    int tz, ty, tx0, ntile, sample;
};

__device__ __forceinline__ bool hot_strip(const HotGeom& hg, HotStrip& sp, int b)
{
    // (a second level with these kernels -- one spilled tile per work item in a 64 KiB block -- was
    // tried and lost to the general level-2 kernel: a full prologue per tile, two workgroups per CU)
    // strips are dealt to the 8 XCDs in contiguous chunks (block b runs on XCD b % 8)
    const int per = (hg.total_strips + 7) >> 3;
    int s = (b & 7) * per + (b >> 3);
    if (s >= hg.total_strips)
        return false;
    sp.sample = s / hg.nstrips;
    s -= sp.sample * hg.nstrips;
    const int sx = s % hg.strips_x;
    s /= hg.strips_x;
    sp.ty = s % hg.tiles[1];
    sp.tz = s / hg.tiles[1];
    sp.tx0 = sx * hg.strip_tiles;
    sp.ntile = min(hg.strip_tiles, hg.tiles[2] - sp.tx0);
    return true;
}

__device__ __forceinline__ void hot_prologue(const HotGeom& hg, const HotStrip& sp, char* smem, int tid,
                                             bool copy_q = true, int nthreads = kBlock)
{
    int* sred = reinterpret_cast<int*>(smem + kOffRed);
    {   // x table: 64 entries x 48 bytes = 768 dwords
        const int* src = reinterpret_cast<const int*>(hg.xt + sp.tx0 * kT);
        int* dst = reinterpret_cast<int*>(smem + kOffTabX);
        const int avail = (hg.out_len[2] - sp.tx0 * kT) * 12;
        for (int e = tid; e < kStrip * kT * 12; e += nthreads)
            dst[e] = e < avail ? src[e] : 0;
    }
    if (copy_q && tid < 256) {   // Q rows: (zi, yy) -> (oz, oy); 4 threads per row, 16 bytes at a time
        const int row16 = 2 * hg.ncpx;                   // 16-byte pieces per row (32 bytes per column)
        const int r = tid >> 2;
        const int oz = min(sp.tz * kT + (r >> 3), hg.out_len[0] - 1);
        const int oy = min(sp.ty * kT + (r & 7), hg.out_len[1] - 1);
        // (wide control grids: Q is laid out per x-strip, hg.ncpx columns each -- TileGeom::q_win)
        const long long qrow_id = hg.q_strips > 1
                                      ? ((long long)oz * hg.out_len[1] + oy) * hg.q_strips + sp.tx0 / hg.strip_tiles
                                      : (long long)oz * hg.out_len[1] + oy;
        const double2* src = reinterpret_cast<const double2*>(
            hg.q + sp.sample * hg.q_bstride + qrow_id * (4 * hg.ncpx));
        double2* dst = reinterpret_cast<double2*>(smem + kOffQ) + r * row16;
        for (int k = tid & 3; k < row16; k += 4)
            dst[k] = src[k];
    }
    if (tid < 24) {
        const int k = tid & 7;
        sred[tid] = k < 3 ? 0x7fffffff : (int)0x80000000;
    }
    if (tid >= 128 && tid < 128 + 12) {
        HotParams* hp = reinterpret_cast<HotParams*>(smem + kOffHot);
        const int k = tid - 128;
        hp->affine[k] = hg.affine[k];
        if (k < 3) {
            hp->offd[k] = (double)hg.off[k];
            hp->last[k] = (double)(hg.in_len[k] - 1);
            hp->period[k] = hg.period[k];
            hp->inv_period[k] = hg.inv_period[k];
        }
        if (k < 8) {
            hp->step_len[k] = hg.step_len[k];
            hp->in_step_stride[k] = hg.vol_step[k];
            hp->out_step_stride[k] = hg.img_step[k];
        }
        if (k == 0)
            hp->nstep = hg.nstep;
    }
    __syncthreads();
}

// Phase A for one voxel (deform.c:649-824): displacement from the lane's Q row, (affine), + offset,
// window start and fractional offsets.  `b[h]` = output index + crop offset along axis h (no affine)
// or 0 (affine: the real base is in `P`).  Returns true when the voxel maps to the constant.
template <int ORDER, bool AFFINE>
__device__ __forceinline__ bool hot_coords(const HotGeom& hg, const HotParams* hp, const char* qrow,
                                           const double (&tw)[4], const int (&tib)[4], const int (&b)[3],
                                           const double (&P)[3], int* start, float* frac)
{
    double d[3];
    {
        // 4 control columns x 3 components: ds_read_b128 (components 0, 1) + ds_read_b64 (2) per column
        double2 q01[4];
        double q2[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            q01[l] = *reinterpret_cast<const double2*>(qrow + tib[l]);
            q2[l] = *reinterpret_cast<const double*>(qrow + tib[l] + 16);
        }
        d[0] = tw[0] * q01[0].x;
        d[1] = tw[0] * q01[0].y;
        d[2] = tw[0] * q2[0];
#pragma unroll
        for (int l = 1; l < 4; ++l) {
            d[0] = fma(tw[l], q01[l].x, d[0]);
            d[1] = fma(tw[l], q01[l].y, d[1]);
            d[2] = fma(tw[l], q2[l], d[2]);
        }
    }
    int ci[3];
    bool inr[3];
#pragma unroll
    for (int h = 0; h < 3; ++h)
        inr[h] = coord_axis_fast<ORDER, float>(AFFINE ? P[h] + d[h] : d[h], AFFINE ? 0 : b[h], hg.in_len[h],
                                               ci[h], frac[h]);
    bool cst = false;
    if (!(inr[0] && inr[1] && inr[2])) {
        // one divergent region: the axes along which the source point left the array
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            if (!inr[h])
                cst = coord_axis_mapped<ORDER, float>(AFFINE ? P[h] + d[h] : (double)b[h] + d[h], hg.in_len[h],
                                                      hg.mode, hp->period[h], hp->inv_period[h], ci[h],
                                                      frac[h]) || cst;
        }
    }
#pragma unroll
    for (int h = 0; h < 3; ++h)
        start[h] = cst ? 0 : ci[h] - ORDER / 2;
    return cst;
}

__device__ __forceinline__ void hot_step_offsets(const HotParams* hp, long long ss, long long& vol_off,
                                                 long long& img_off)
{
    vol_off = 0;
    img_off = 0;
    long long r = ss;
    const int nstep = hp->nstep;
    for (int l = 0; l < nstep; ++l) {
        const long long len = hp->step_len[l];
        const long long q = r / len;
        const long long c = r - q * len;
        vol_off += hp->in_step_stride[l] * c;
        img_off += hp->out_step_stride[l] * c;
        r = q;
    }
}

}  // namespace
}  // namespace tile
}  // namespace ed
