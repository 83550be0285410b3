// deform_hot.hip -- K1 (forward gather) and K2 (gradient scatter-add) for the benchmark case: float32
// volumes, 3 deformed axes, unit stride along x on both sides, spline orders 1-5.  Same algorithm and
// LDS tiling as the general kernels of deform_tile.hip (which keep serving float64, strided layouts and
// the tiles these kernels hand to the spill list), rebuilt around the instruction costs measured on
// MI355X (profiles/r02_ubench_valu.txt): fp32 FMA / integer add issue in 2 cycles per wave, everything
// else -- fp64, conversions, DPP, v_readlane, 3-operand integer ops -- in 4, so the per-voxel work is
// cut where the 4-cycle instructions were:
//
//   * the kernel argument block is small (HotGeom) and the rarely used uniform values live in LDS:
//     the general kernel kept ~100 SGPRs of arguments live and spilled them through v_writelane /
//     v_readlane (a fifth of its VALU instructions);
//   * coordinates: without an affine map floor / fraction are taken of the DISPLACEMENT alone and the
//     output index is added as an integer (exact; three int->fp64 conversions and fp64 adds less per
//     voxel); the boundary map stays one divergent region for the few lanes that leave the array;
//   * Q rows are [control column][component padded to 4]: a voxel's 12 fp64 taps are 4 ds_read_b128 +
//     4 ds_read_b64 off 4 addresses, which depend on the lane only (computed once per tile);
//   * staging of interior tiles walks rows with incremental (z, y) counters instead of two divisions
//     and two mirror maps per 16-byte chunk;
//   * the gather is instantiated per row pitch, so the 16 row offsets of a voxel are immediates;
//   * bounding-box reductions finish inside the wave with DPP row_bcast steps (no v_readlane).
//
// Reference: the per-voxel pipeline of DeformGrid, deform.c:649-1001 (see deform_tile.hip's header
// for the phase-by-phase mapping).
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

// 64-tap (order 3) separable gather of one voxel from the staged box; PITCH is a template argument so
// that the row offsets are immediates.  `bp` points at tap (0, 0, 0) in the copy whose shift matches
// the parity of the window's x start: every x-run is a sequence of aligned ds_read_b64.
template <int ORDER, int PITCH>
__device__ __forceinline__ float hot_gather(const float* bp, int plane, const float* w0, const float* w1,
                                            const float* w2)
{
    constexpr int NT = ORDER + 1;
    constexpr int NTX = NT + (NT & 1);
    float a0 = 0.f;
#pragma unroll
    for (int l0 = 0; l0 < NT; ++l0) {
        const float* pp = bp + l0 * plane;
        float a1 = 0.f;
#pragma unroll
        for (int l1 = 0; l1 < NT; ++l1) {
            const float* rp = pp + l1 * PITCH;
            float a2 = 0.f;
#pragma unroll
            for (int l2 = 0; l2 < NTX; l2 += 2) {
                const float2 pr = *reinterpret_cast<const float2*>(rp + l2);
                a2 = fmaf(w2[l2], pr.x, a2);
                a2 = fmaf(w2[l2 + 1], pr.y, a2);
            }
            a1 = fmaf(w1[l1], a2, a1);
        }
        a0 = fmaf(w0[l0], a1, a0);
    }
    return a0;
}

// ================================================================================================
// K1: forward
// ================================================================================================
// async global -> LDS copy of 16 bytes per lane (LDS-DMA): the wave's 64 lanes fill 1 KiB of LDS
// starting at the wave-uniform `lds`, in lane order; the global address is per lane.  No VGPR, no
// ds_write, and the wave keeps running: the data is ordered for readers by s_waitcnt vmcnt + barrier.
__device__ __forceinline__ void glds16(const float* g, float* lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// phases A + B of one tile for this lane's two voxels: coordinates, then the bounding box of the
// tile's tap windows reduced into the LDS slots `red` (min x3, max x3)
template <int ORDER, bool AFFINE, int ABL, int NV>
__device__ __forceinline__ void hot_tile_coords(const HotGeom& hg, const HotParams* hp, const AxTab* tabx,
                                                int* red, const char* const (&qrow)[NV], const int (&oz)[NV],
                                                int oy, int ox0, int xx, int lane, const bool (&vzy)[NV],
                                                const double (&Pzy)[3][NV], int (&start)[NV][3],
                                                float (&frac)[NV][3], bool (&valid)[NV], bool (&constant)[NV])
{
    constexpr int kPadX = (ORDER + 1) & 1;
    const int ox = ox0 + xx;
    const bool vx = ox < hg.out_len[2];
    double tw[4];
    int tib[4];
    {
        const AxTab& t = tabx[xx];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            tw[l] = t.w[l];
            tib[l] = t.idx[l] * 8;          // idx counts doubles of a Q row: byte offset
        }
    }
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int b[3] = {oz[i] + hg.off[0], oy + hg.off[1], ox + hg.off[2]};
        double P[3] = {0.0, 0.0, 0.0};
        if (AFFINE) {
#pragma unroll
            for (int h = 0; h < 3; ++h)
                P[h] = fma(hp->affine[h * 4 + 2], (double)ox, Pzy[h][i]);
        }
        if (ABL & 4) {
            constant[i] = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                start[i][h] = min(max(b[h] - 1, 0), hg.in_len[h] - 4);
                frac[i][h] = 0.5f + (float)tw[0] * 1e-30f + (float)tib[0] * 1e-30f;
            }
        } else
        constant[i] = hot_coords<ORDER, AFFINE>(hg, hp, qrow[i], tw, tib, b, P, start[i], frac[i]);
        valid[i] = vzy[i] && vx;
        if (valid[i] && !constant[i]) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                lo[h] = min(lo[h], start[i][h]);
                hi[h] = max(hi[h], start[i][h] + ORDER + (h == 2 ? kPadX : 0));
            }
        }
    }
    if (!(ABL & 8))
        box_reduce_to_lds(red, lane, lo, hi);
}

// self_serve: the tiles of a strip whose source box did not fit the LDS box, gathered straight from global
// memory behind the strip loop -- a rare path (2 tiles of 32768 on the benchmark volume) that saves the call
// the two launches of the spill levels.  Coordinates from the strip's tables in LDS with hot_coords, taps
// mirror-mapped per axis (deform.c:791-813), accumulation x, y, z as chains of fused multiply-adds from zero:
// the bits every other level gives.  Not inlined: called where nothing of the tile loop is live any more
// (inlined into the loop an earlier form cost K1 its register allocation, profiles/r03_bench_misc.txt).
template <int ORDER, bool AFFINE>
__device__ __forceinline__ void hot_fwd_unfit(const HotGeom& hg, const HotStrip& sp, char* smem, unsigned unfit, int io16)
{
    constexpr int NT = ORDER + 1;
    const AxTab* tabx = reinterpret_cast<const AxTab*>(smem + kOffTabX);
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem + kOffHot);
    const int tid = threadIdx.x;
    const int yy = (tid >> 3) & 7, xx = tid & 7, zq = tid >> 6;
    const float* __restrict__ vol = hg.vol_r + sp.sample * hg.vol_bstride;
    float* img = hg.img_w + sp.sample * hg.img_bstride;
    const int oy = sp.ty * kT + yy;
    for (int ti = 0; ti < sp.ntile; ++ti) {
        if (!((unfit >> ti) & 1u))
            continue;
        const int ox = (sp.tx0 + ti) * kT + xx;
        double tw[4];
        int tib[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            tw[l] = tabx[ti * kT + xx].w[l];
            tib[l] = tabx[ti * kT + xx].idx[l] * 8;
        }
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            const int zi = zq + 4 * i;
            const int oz = sp.tz * kT + zi;
            if (oz >= hg.out_len[0] || oy >= hg.out_len[1] || ox >= hg.out_len[2])
                continue;
            const char* qrow = smem + kOffQ + (zi * kT + yy) * (32 * hg.ncpx);
            const int b[3] = {oz + hg.off[0], oy + hg.off[1], ox + hg.off[2]};
            double P[3] = {0.0, 0.0, 0.0};
            if (AFFINE) {
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    P[h] = fma(hp->affine[h * 4 + 2], (double)ox,
                               fma(hp->affine[h * 4 + 0], (double)oz,
                                   fma(hp->affine[h * 4 + 1], (double)oy, hp->affine[h * 4 + 3] + hp->offd[h])));
            }
            int st[3];
            float fr[3];
            const bool cst = hot_coords<ORDER, AFFINE>(hg, hp, qrow, tw, tib, b, P, st, fr);
            int tap[3][NT];
            float w[3][NT];
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                weights_from_frac<float, ORDER>(fr[h], w[h]);
                const int stride = h == 0 ? hg.vol_sz : (h == 1 ? hg.vol_sy : 1);
#pragma unroll
                for (int l = 0; l < NT; ++l)
                    tap[h][l] = mirror_i32(st[h] + l, hg.in_len[h]) * stride;
            }
            const int obase = oz * hg.img_sz + oy * hg.img_sy + ox;
            for (long long ss = 0; ss < hg.nsteps; ++ss) {
                long long vol_off = 0, img_off = 0;
                if (hg.nstep)
                    hot_step_offsets(hp, ss, vol_off, img_off);
                float val = hg.cval;
                if (!cst) {
                    const float* src = vol + vol_off;
                    float a0 = 0.f;
#pragma unroll
                    for (int l0 = 0; l0 < NT; ++l0) {
                        float a1 = 0.f;
#pragma unroll
                        for (int l1 = 0; l1 < NT; ++l1) {
                            const float* p1 = src + (tap[0][l0] + tap[1][l1]);
                            float a2 = 0.f;
#pragma unroll
                            for (int l2 = 0; l2 < NT; ++l2)
                                a2 = fmaf(w[2][l2], p1[tap[2][l2]], a2);
                            a1 = fmaf(w[1][l1], a2, a1);
                        }
                        a0 = fmaf(w[0][l0], a1, a0);
                    }
                    val = a0;
                }
                store_out(img, img_off + obase, val, io16);
            }
        }
    }
}

// ABL: compile-time ablation switches for profiling (0 in production; EDHIP_HOT_ABL selects one of
// the instantiated values for order 3): 2 skip the gather, 4 skip the coordinates, 8 skip the
// bounding-box reduction (analytic box), 32 skip staging, 64 skip the output store
//
// Tile loop, software-pipelined: once the box of tile t is known its staging copies are issued as
// asynchronous LDS-DMA, and the coordinates + bounding box of tile t + 1 are computed while they are
// in flight; the gather of tile t follows the barrier that retires the copies.
// IO16: the output is stored as 16-bit floats (HotGeom::io16 says which); instantiated for orders 1-3 only
template <int ORDER, bool AFFINE, int ABL = 0, int NTH = kBlock, int WAVES = ((ABL & 2048) ? 5 : 4), bool REC_ONLY = false,
          bool IO16 = false>
__global__ __launch_bounds__(NTH, WAVES) void hot_fwd_kernel(const HotGeom hg)
{
    const int io16 = IO16 ? hg.io16 : 0;
    // NTH = 256: two voxels per lane (z = wave, wave + 4); NTH = 512: one voxel per lane, eight waves
    constexpr int NV = 512 / NTH;
    constexpr int NW = NTH / 64;
    // coordinates of tile t + 1 computed under tile t's copies (orders 4 / 5: the extra live state spills)
    constexpr bool PIPE = !(ABL & 4096) && ORDER <= 3;
    constexpr bool QGLOBAL = (ABL & 1024) != 0; // experiment: Q rows read from global memory (L1), not LDS
    constexpr int NT = ORDER + 1;
    constexpr int kPadX = NT & 1;          // even orders read one zero-weight padding tap
    constexpr int NTX = NT + kPadX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    HotStrip sp;
    if (!hot_strip(hg, sp, blockIdx.x))
        return;
    // records-only launch (first half of a gradient call): nothing to do for a sample whose records were
    // made by a forward call from these very displacement values (flag written by the tables kernel)
    if (REC_ONLY && hg.rec_valid && hg.rec_valid[sp.sample])
        return;
    // (per-workgroup issue priorities (s_setprio) and a staggered start of the workgroups of a CU, to
    // push co-resident workgroups into complementary phases, were tried: no change)
    hot_prologue(hg, sp, smem, threadIdx.x, !QGLOBAL, NTH);
    if (ED_DBG(hg.dbg, 8192))
        return;       // experiment: launch + prologue only

    const AxTab* tabx = reinterpret_cast<const AxTab*>(smem + kOffTabX);
    int* sred = reinterpret_cast<int*>(smem + kOffRed);
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem + kOffHot);
    float* box0 = reinterpret_cast<float*>(smem + (QGLOBAL ? kOffQ : hg.off_box));
    float* box1 = box0 + hg.box_cap;      // cap = 56 (mod 64): the two copies sit on disjoint banks

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int yy = lane >> 3, xx = lane & 7;
    const float* __restrict__ vol = hg.vol_r + sp.sample * hg.vol_bstride;
    float* img = hg.img_w + sp.sample * hg.img_bstride;

    // per-lane values that stay fixed along the strip
    const int oy = sp.ty * kT + yy;
    const char* qrow[NV];
    int oz[NV], obase[NV];
    bool vzy[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int zi = wave + NW * i;
        oz[i] = sp.tz * kT + zi;
        if (QGLOBAL)
            qrow[i] = reinterpret_cast<const char*>(
                hg.q + sp.sample * hg.q_bstride +
                ((long long)min(oz[i], hg.out_len[0] - 1) * hg.out_len[1] + min(oy, hg.out_len[1] - 1)) * (4 * hg.ncpx));
        else
            qrow[i] = smem + kOffQ + (zi * kT + yy) * (32 * hg.ncpx);
        vzy[i] = oz[i] < hg.out_len[0] && oy < hg.out_len[1];
        obase[i] = oz[i] * hg.img_sz + oy * hg.img_sy + sp.tx0 * kT + xx;
    }
    double Pzy[3][NV];     // affine: A[h][0] oz + A[h][1] oy + A[h][3] + off_h
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            Pzy[h][i] = 0.0;
    if (AFFINE) {
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
            for (int i = 0; i < NV; ++i)
                Pzy[h][i] = fma(hp->affine[h * 4 + 0], (double)oz[i],
                                fma(hp->affine[h * 4 + 1], (double)oy, hp->affine[h * 4 + 3] + hp->offd[h]));
    }

    int start[NV][3];
    float frac[NV][3];
    bool valid[NV], constant[NV];
    unsigned unfit = 0;       // self_serve: tiles of this strip whose box did not fit
    if (PIPE)
        hot_tile_coords<ORDER, AFFINE, ABL, NV>(hg, hp, tabx, sred, qrow, oz, oy, sp.tx0 * kT, xx, lane, vzy, Pzy,
                                            start, frac, valid, constant);

    for (int ti = 0; ti < sp.ntile; ++ti) {
        int* red = sred + (ti % 3) * 8;
        if (!PIPE)
            hot_tile_coords<ORDER, AFFINE, ABL, NV>(hg, hp, tabx + ti * kT, red, qrow, oz, oy, (sp.tx0 + ti) * kT, xx,
                                                lane, vzy, Pzy, start, frac, valid, constant);
        lds_atomics_done();
        if (ED_DBG(hg.dbg, 2048)) __syncthreads(); else lds_barrier();   // B1: box known; every gather of the previous tile is done
        int b0[3] = {red[0], red[1], red[2]};
        int ext[3] = {red[3] - red[0] + 1, red[4] - red[1] + 1, red[5] - red[2] + 1};
        bool any = red[3] >= red[0];
        if (hg.boxes && tid < 6)       // EDHIP_FLAG_KEEP_BOXES: the box goes to the gradient call too
            hg.boxes[(size_t)(sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 + ti) * 8 +
                     tid] = red[tid] - ((tid == 5 && any) ? kPadX : 0);      // (without the forward gather's padding tap)
        if (hg.rec) {
            // coordinate records for the gradient kernel (hot_grad2_kernel): window start relative to this
            // tile's box + the three fractions, for every voxel of the output -- also of a tile that is
            // handed to the spill list below (the gradient's tiles are twice as long and hold other boxes)
            typedef float f4_t __attribute__((ext_vector_type(4)));
            const int ox = (sp.tx0 + ti) * kT + xx;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (valid[i]) {
                    unsigned w = kRecDead;
                    if (!constant[i])
                        w = (unsigned)min(start[i][0] - b0[0], 255) | ((unsigned)min(start[i][1] - b0[1], 255) << 8) |
                            ((unsigned)min(start[i][2] - b0[2], 255) << 16);
                    const f4_t r = {frac[i][0], frac[i][1], frac[i][2], __uint_as_float(w)};
                    f4_t* dst = reinterpret_cast<f4_t*>(hg.rec) +
                                (sp.sample * hg.rec_bstride + ((long long)oz[i] * hg.out_len[1] + oy) * hg.out_len[2] + ox);
                    __builtin_nontemporal_store(r, dst);
                }
            }
        }
        if (ABL & 8) {        // (only meaningful together with ABL & 4: identity coordinates)
            any = true;
            b0[0] = min(max(sp.tz * kT + hg.off[0] - 1, 0), hg.in_len[0] - 4);
            b0[1] = min(max(sp.ty * kT + hg.off[1] - 1, 0), hg.in_len[1] - 4);
            b0[2] = min(max((sp.tx0 + ti) * kT + hg.off[2] - 1, 0), hg.in_len[2] - 4);
            ext[0] = ext[1] = ext[2] = kT + 3;
        }
        // re-arm the buffer of tile ti + 2 (three buffers in rotation: tile ti + 1 reduces into its
        // buffer during this iteration, tile ti's is being read)
        if (tid < 6)
            sred[((ti + 2) % 3) * 8 + tid] = tid < 3 ? 0x7fffffff : (int)0x80000000;
        // row pitch 16 * odd: four consecutive rows sit on four disjoint groups of 16 banks
        const int pitch = ext[2] <= 16 ? 16 : (ext[2] <= 48 ? 48 : 0);
        // (ABL & 256: rows per plane padded to a multiple of 4, so that the bank of a tap depends on
        // its (row mod 4, x) only -- 14 % fewer bank-conflict cycles, 8 us faster, but more tiles
        // overflow the box and the call as a whole loses: not used)
        const int by = (ABL & 256) ? (ext[1] + 3) & ~3 : ext[1];
        const int nrows = ext[0] * by;
        const bool fits = pitch > 0 && nrows * pitch <= hg.box_cap;
        const bool staged = any && fits;
        if (!REC_ONLY && any && hg.hint && tid == 0 && !(pitch > 0 && nrows * pitch <= hg.small_cap))
            atomicAdd(hg.hint, 1);         // spill feedback: would not fit the standard box
        if (!REC_ONLY && any && !fits && hg.self_serve)
            unfit |= 1u << ti;                 // served below, behind the strip loop
        else if (!REC_ONLY && any && !fits && tid == 0) {    // hand the whole tile to the general kernels
            const int slot = atomicAdd(&hg.spill[0], 1);
            hg.spill[1 + slot] = sp.sample * hg.ntiles +
                                 (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 + ti;
        }
        // the box and the padded row behind it lie inside the volume: no mirror map while staging
        const bool interior = b0[0] >= 0 && b0[0] + ext[0] <= hg.in_len[0] && b0[1] >= 0 &&
                              b0[1] + ext[1] <= hg.in_len[1] && b0[2] >= 0 &&
                              b0[2] + pitch + 1 <= hg.in_len[2];
        const bool x_inside = b0[2] >= 0 && b0[2] + ext[2] <= hg.in_len[2];

        // ---- phase C: stage the source box into LDS (two copies, the second shifted by one) ----------
        auto stage = [&](const float* src) {
            if (interior && pitch == 16 && (ABL & 16)) {
                // experiment (ABL & 16): one 16-byte load per chunk, the shifted copy built in
                // registers -- element 4 of the shifted chunk is the neighbouring lane's first element
                // (the four lanes of a row are a DPP quad).  Half the loads of fetching both copies,
                // two ds_write_b128 instead of two LDS-DMA copies: 248 us against 240 us (not used).
                const int q = tid & 3;
                const float inv_by = 1.0f / (float)by;
                for (int r = tid >> 2; r < nrows; r += NTH / 4) {
                    const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                    const float* g = src + ((b0[0] + zr) * hg.vol_sz + (b0[1] + yr) * hg.vol_sy + b0[2] + 4 * q);
                    const F4u v0 = *reinterpret_cast<const F4u*>(g);
                    const float nx = __int_as_float(__builtin_amdgcn_update_dpp(
                        0, __float_as_int(v0.x), 0xF9 /* quad_perm:[1,2,3,3] */, 0xf, 0xf, false));
                    float* lp = box0 + r * 16 + 4 * q;
                    *reinterpret_cast<float4*>(lp) = make_float4(v0.x, v0.y, v0.z, v0.w);
                    *reinterpret_cast<float4*>(lp + hg.box_cap) = make_float4(v0.y, v0.z, v0.w, nx);
                }
            } else if (interior) {
                // LDS-DMA: one wave-instruction fills 1 KiB = RW consecutive box rows (16 rows of 64
                // bytes, or 5 rows of 192 bytes with lanes 60-63 idle); lane -> (row, 16-byte chunk)
                const int cpr = pitch >> 2;
                const int RW = pitch == 16 ? 16 : 5;
                const int lrow = pitch == 16 ? lane >> 2 : (lane * 21846) >> 18;      // lane / 12
                const int q = lane - lrow * cpr;
                const float inv_by = 1.0f / (float)by;
                for (int r0 = wave * RW; r0 < nrows; r0 += NW * RW) {
                    const int r = r0 + lrow;
                    const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                    if (lrow < RW && r < nrows && yr < ext[1]) {
                        const float* g = src + ((b0[0] + zr) * hg.vol_sz + (b0[1] + yr) * hg.vol_sy + b0[2] + 4 * q);
                        glds16(g, box0 + r0 * pitch);
                        glds16(g + 1, box1 + r0 * pitch);
                    }
                }
            } else {
                // edge tile: every box index goes through the mirror map, as the reference does
                // with the taps of a window that sticks out (deform.c:791-813)
                const float inv_by = 1.0f / (float)by;
                const int sub = tid & 7;
                for (int r = tid >> 3; r < nrows; r += NTH / 8) {
                    const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                    if (yr >= ext[1])
                        continue;            // padding row of the plane
                    const int zs = mirror_i32(b0[0] + zr, hg.in_len[0]);
                    const int ys = mirror_i32(b0[1] + yr, hg.in_len[1]);
                    const float* rowp = src + (zs * hg.vol_sz + ys * hg.vol_sy);
                    float* d0 = box0 + r * pitch;
                    float* d1 = box1 + r * pitch;
                    for (int xi = sub; xi < ext[2]; xi += 8) {
                        const int xs = x_inside ? b0[2] + xi : mirror_i32(b0[2] + xi, hg.in_len[2]);
                        const float val = rowp[xs];
                        d0[xi] = val;
                        if (xi > 0)
                            d1[xi - 1] = val;
                    }
                }
            }
        };
        long long vol_off = 0, img_off = 0;
        if (hg.nstep)
            hot_step_offsets(hp, 0, vol_off, img_off);
        if (!REC_ONLY && staged && !(ABL & 32))
            stage(vol + vol_off);

        // ---- phases A + B of the NEXT tile, while the copies are in flight ------------------------
        int nstart[NV][3];
        float nfrac[NV][3];
        bool nvalid[NV], nconstant[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i)
            nvalid[i] = nconstant[i] = false;
        if (PIPE && ti + 1 < sp.ntile)
            hot_tile_coords<ORDER, AFFINE, ABL, NV>(hg, hp, tabx + (ti + 1) * kT, sred + ((ti + 1) % 3) * 8, qrow, oz,
                                                oy, (sp.tx0 + ti + 1) * kT, xx, lane, vzy, Pzy, nstart, nfrac,
                                                nvalid, nconstant);

        if (!REC_ONLY && !(any && !fits)) {
            for (long long ss = 0; ss < hg.nsteps; ++ss) {
                if (ss > 0) {
                    hot_step_offsets(hp, ss, vol_off, img_off);
                    if (staged && !(ABL & 32)) {
                        __syncthreads();     // previous step's gathers are done with the box
                        stage(vol + vol_off);
                    }
                }
                if (staged && (!(ABL & 32) || (ABL & 512)))
                    __syncthreads();         // B2: retires this wave's copies (vmcnt) and everyone's

                // ---- phase D: gather -------------------------------------------------------------
                // (both voxels in one branch-free block, for the scheduler to overlap one voxel's LDS
                // reads with the other's arithmetic, was tried: 128 VGPRs + 192 bytes of scratch, 407 us)
                const int plane = by * pitch;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (!IO16 && !valid[i])
                        continue;
                    float val = 0.f;
                    if (IO16 && !valid[i]) {
                        ;
                    } else if (constant[i]) {
                        val = hg.cval;
                    } else if (ABL & 2) {
                        val = frac[i][0] + frac[i][1] + frac[i][2] + (float)start[i][0] + (float)start[i][1] + (float)start[i][2];
                    } else {
                        float w0[NT], w1[NT], w2[NTX];
                        weights_from_frac<float, ORDER>(frac[i][0], w0);
                        weights_from_frac<float, ORDER>(frac[i][1], w1);
                        weights_from_frac<float, ORDER>(frac[i][2], w2);
                        if (kPadX)
                            w2[NT] = 0.f;
                        const int rz = start[i][0] - b0[0], ry = start[i][1] - b0[1], rx = start[i][2] - b0[2];
                        // aligned pairs from the copy whose shift matches the parity of rx
                        const float* bp = ((rx & 1) ? box1 - 1 : box0) + ((rz * by + ry) * pitch + rx);
                        val = pitch == 16 ? hot_gather<ORDER, 16>(bp, plane, w0, w1, w2)
                                          : hot_gather<ORDER, 48>(bp, plane, w0, w1, w2);
                    }
                    // streaming store (a tile writes 32-byte row segments; see deform_tile.hip)
                    if constexpr (IO16) {
                        // (cached 2-byte stores: a tile writes 16-byte row segments, half a 32-byte sector -- streamed
                        // past the L2 they cost K1 17 us on the 256^3 benchmark; pairing x-neighbours' lanes into
                        // 32-bit stores cost 12 spilled registers and more: 254 us against 248)
                        if (valid[i])
                            reinterpret_cast<unsigned short*>(img)[img_off + obase[i] + ti * kT] = (unsigned short)narrow16(val, io16);
                    } else if (!(ABL & 64) || val == -12345.678f)
                        store_out(img, img_off + obase[i] + ti * kT, val, io16);
                }
            }
        }
        if (PIPE) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                valid[i] = nvalid[i];
                constant[i] = nconstant[i];
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    start[i][h] = nstart[i][h];
                    frac[i][h] = nfrac[i][h];
                }
            }
        }
    }
    if (!REC_ONLY && unfit)
        hot_fwd_unfit<ORDER, AFFINE>(hg, sp, smem, unfit, io16);
}

// ================================================================================================
// K2: gradient.  Tiles of 8 (z) x 8 (y) x 16 (x) voxels, four voxels per lane; taps are scattered
// into fixed-point LDS cells with integer atomics and flushed with one float atomic per touched
// source element (see deform_tile.hip for the scale's no-overflow bound).
// ================================================================================================
// IO16: dY is stored as 16-bit floats (HotGeom::io16 says which); instantiated for orders 1-3 only
template <int ORDER, bool AFFINE, int GRAD_WAVES, int TX, int NGRP = 1, bool IO16 = false>
__global__ __launch_bounds__(kBlock * NGRP, GRAD_WAVES) void hot_grad_kernel(const HotGeom hg)
{
    const int io16 = IO16 ? hg.io16 : 0;
    constexpr int NT = ORDER + 1;
    constexpr int NV = TX / 4, ZSTEP = 8 / NV;       // TX 16: 4 voxels per lane; TX 8: 2
    extern __shared__ __attribute__((aligned(16))) char smem0[];
    // NGRP 2 (experiment): two groups of four waves, each with a strip and an LDS region of its
    // own, one barrier interval apart -- group 1 scatters (LDS atomics) while group 0 flushes and
    // computes coordinates (VALU), instead of four workgroups marching through the phases together
    const int grp = NGRP == 2 ? (int)(threadIdx.x >> 8) : 0;
    const int tid = threadIdx.x & (kBlock - 1);
    char* smem = smem0 + grp * hg.lds_grp;
    int phase = 0;
    HotStrip sp;
    const int bidx = NGRP == 2 ? (int)(((blockIdx.x >> 3) * 2 + grp) * 8 + (blockIdx.x & 7)) : (int)blockIdx.x;
    if (!hot_strip(hg, sp, bidx))
        return;
    // the accumulator cells start at zero and every flush leaves the cells it read at zero again
    for (int e = tid * 4; e < hg.box_cap; e += kBlock * 4)
        *reinterpret_cast<int4*>(smem + hg.off_box + e * 4) = make_int4(0, 0, 0, 0);
    hot_prologue(hg, sp, smem, tid);
    if (ED_DBG(hg.dbg, 8192))
        return;       // experiment: launch + prologue only
    if (NGRP == 2 && grp == 1)
        __syncthreads();

    const AxTab* tabx = reinterpret_cast<const AxTab*>(smem + kOffTabX);
    int* sred = reinterpret_cast<int*>(smem + kOffRed);
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem + kOffHot);
    int* box = reinterpret_cast<int*>(smem + hg.off_box);

    const int lane = tid & 63;
    const int wave = tid >> 6;
    // lane -> voxel: the 16 lanes that go through the LDS together hold voxels two apart along x
    // and y (even / odd x, rows y and y + 2).  Neighbours along an axis share a window start where
    // the deformation compresses, and two lanes adding into one cell serialise the atomic: 326 ->
    // 307 us.  (All 64 lanes two apart along x, y and z: no further gain.)
    const int zq = tid / (TX * 8);
    const int xx = TX == 16 ? 2 * (tid & 7) + ((tid >> 4) & 1) : (tid & (TX - 1));
    const int yy = TX == 16 ? 4 * ((tid >> 6) & 1) + ((tid >> 5) & 1) + 2 * ((tid >> 3) & 1) : ((tid / TX) & 7);
    const int ntile = (sp.ntile * kT + TX - 1) / TX;
    float* dx = hg.vol_w + sp.sample * hg.vol_bstride;
    const float* __restrict__ dy = hg.img_r + sp.sample * hg.img_bstride;

    const int oy = sp.ty * kT + yy;
    const char* qrow0 = smem + kOffQ + (zq * kT + yy) * (32 * hg.ncpx);      // voxel i: + i * ZSTEP rows of 8
    const int qstep = ZSTEP * kT * 32 * hg.ncpx;
    const int oz0 = sp.tz * kT + zq;
    const bool vy = oy < hg.out_len[1];

    // dY of a tile's first step is loaded one tile ahead: with the forward call's boxes there is no
    // box pass left to hide its HBM round trip under.  (IO16: the 16 raw bits travel in the register and are widened
    // where the value is used -- a conversion at the load would wait for it right there.)
    auto load_raw = [&](long long off) -> float {
        if constexpr (IO16)
            return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(dy)[off]);
        else
            return dy[off];
    };
    float gnext[NV];
    {
        const int ox = sp.tx0 * kT + xx;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            gnext[i] = (vy && ox < hg.out_len[2] && oz0 + ZSTEP * i < hg.out_len[0])
                           ? load_raw((oz0 + ZSTEP * i) * hg.img_sz + oy * hg.img_sy + ox) : 0.f;
    }
    // half: -1 = the whole TX-wide tile; 0 / 1 = an oversize tile taken again as its x-halves (below)
    int half = -1, half_next = -1;
    for (int ti = 0; ti < ntile; half = half_next, half_next = half == 0 ? 1 : -1, ti += half < 0 ? 1 : 0) {
        int* red = sred + (ti % 3) * 8;
        const int ox = sp.tx0 * kT + ti * TX + xx;
        const bool vx = ox < hg.out_len[2];
        const int ooff0 = oz0 * hg.img_sz + oy * hg.img_sy + ox;
        const int ostep = ZSTEP * hg.img_sz;

        float gpre[NV];
        if (half >= 0) {
            // (the tile's dY again: gnext already holds the next tile's)
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gpre[i] = (vy && vx && oz0 + ZSTEP * i < hg.out_len[0]) ? load_raw(ooff0 + i * ostep) : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gpre[i] = gnext[i];
        }
        if (half < 0 && ti + 1 < ntile) {
            const bool nvx = ox + TX < hg.out_len[2];
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gnext[i] = (vy && nvx && oz0 + ZSTEP * i < hg.out_len[0]) ? load_raw(ooff0 + TX + i * ostep) : 0.f;
        }
        // the forward call's boxes (EDHIP_FLAG_USE_BOXES): requested here, ahead of the barrier
        const bool given = hg.use_boxes != 0;
        int gb0[3] = {0, 0, 0}, gbhi[3] = {-1, -1, -1};
        if (given) {
            const int t0 = sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 +
                           ti * (TX / kT);
            const int* bx = hg.boxes + (size_t)t0 * 8;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                gb0[h] = uni(bx[h]);
                gbhi[h] = uni(bx[3 + h]);
            }
            if (half < 0) {
#pragma unroll
                for (int k = 1; k < TX / kT; ++k) {
                    if (sp.tx0 + ti * (TX / kT) + k < hg.tiles[2]) {
#pragma unroll
                        for (int h = 0; h < 3; ++h) {
                            gb0[h] = min(gb0[h], uni(bx[k * 8 + h]));
                            gbhi[h] = max(gbhi[h], uni(bx[k * 8 + 3 + h]));
                        }
                    }
                }
            } else if (half == 1) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    gb0[h] = uni(bx[8 + h]);
                    gbhi[h] = uni(bx[8 + 3 + h]);
                }
            }
        }

        double tw[4];
        int tib[4];
        {
            const AxTab& t = tabx[ti * TX + xx];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                tw[l] = t.w[l];
                tib[l] = t.idx[l] * 8;
            }
        }
        double Pxy[3] = {0.0, 0.0, 0.0};      // affine: A[h][1] oy + A[h][2] ox + A[h][3] + off_h
        if (AFFINE) {
#pragma unroll
            for (int h = 0; h < 3; ++h)
                Pxy[h] = fma(hp->affine[h * 4 + 2], (double)ox,
                             fma(hp->affine[h * 4 + 1], (double)oy, hp->affine[h * 4 + 3] + hp->offd[h]));
        }
        // One voxel at a time, nothing kept per voxel: the window starts are computed here for the
        // box and again in the scatter pass (the same instructions on the same inputs), so that the
        // kernel holds one voxel's registers instead of four (it spilled 33-79 VGPRs to scratch,
        // and every reload sat on the critical path of its wave).
        auto voxel = [&](int i, int* start, float* frac) -> bool {
            const int oz = oz0 + ZSTEP * i;
            const int b[3] = {oz + hg.off[0], oy + hg.off[1], ox + hg.off[2]};
            double P[3] = {0.0, 0.0, 0.0};
            if (AFFINE) {
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    P[h] = fma(hp->affine[h * 4 + 0], (double)oz, Pxy[h]);
            }
            const bool cst = hot_coords<ORDER, AFFINE>(hg, hp, qrow0 + i * qstep, tw, tib, b, P, start, frac);
            // constant voxels contribute nothing (:928); in a half pass the other half's lanes sit out
            return vy && vx && oz < hg.out_len[0] && !cst && (half < 0 || (xx >> 3) == half);
        };
        // With the forward call's boxes (EDHIP_FLAG_USE_BOXES) the box pass is skipped: this tile's box
        // is the union of the boxes of the 8-wide forward tiles it covers.
        int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
        if (!given) {
ED_UNROLL(ED_K2_U1)
            for (int i = 0; i < NV; ++i) {
                int start[3];
                float frac[3];
                if (voxel(i, start, frac)) {
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        lo[h] = min(lo[h], start[h]);
                        hi[h] = max(hi[h], start[h] + ORDER);
                    }
                }
            }
        }
        // sum of |dY| over the tile (first step): published with the box, under the same barrier
        float gval[NV];
        {
            float gm = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                gval[i] = IO16 ? widen16(__float_as_uint(gpre[i]), io16) : gpre[i];
                // inf / NaN gradients have no fixed-point scale: left out of the sum, scattered with
                // float atomics below
                gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
            }
            gm = wave_sum(gm);
            if (lane == 0)
                reinterpret_cast<float*>(smem + kOffSum)[(phase & 1) * 4 + wave] = gm;
        }
        if (!given)
            box_reduce_to_lds(red, lane, lo, hi);
        lds_barrier();     // B1: box and sum known; the previous tile's flush is done (cells back at zero)
        // wave-uniform box: kept in SGPRs
        int b0[3], bhi[3];
        if (given) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                b0[h] = gb0[h];
                bhi[h] = gbhi[h];
            }
        } else {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                b0[h] = uni(red[h]);
                bhi[h] = uni(red[3 + h]);
            }
        }
        bool any = bhi[0] >= b0[0] && bhi[1] >= b0[1] && bhi[2] >= b0[2];
        if (given && !any) {
            // an empty box may be a stale one: keep going with zero cells -- every live voxel then
            // fails the window test below and is scattered directly
            bhi[0] = b0[0] - 1;
            bhi[1] = b0[1] - 1;
            bhi[2] = b0[2] - 1;
            any = true;
        }
        const int ext[3] = {bhi[0] - b0[0] + 1, bhi[1] - b0[1] + 1, bhi[2] - b0[2] + 1};
        if (tid < 6 && !given)
            sred[((ti + 2) % 3) * 8 + tid] = tid < 3 ? 0x7fffffff : (int)0x80000000;
        if (!any)
            continue;      // nothing to scatter (uniform)
        // 16 lanes of a row hit 16 consecutive cells; pitch 8 * odd keeps neighbouring rows apart
        int pitch = ext[2] <= 8 ? 8 : (ext[2] <= 24 ? 24 : (ext[2] <= 40 ? 40 : (ext[2] <= 56 ? 56 : 0)));
        if (ED_DBG(hg.dbg, 1024))      // experiment: row offsets of 16 banks (mod 32): 34 % fewer conflict
            pitch = ext[2] <= 16 ? 16 : (ext[2] <= 48 ? 48 : 0);     // cycles, but the box doubles
        if (given && ((unsigned)ext[0] > 4096u || (unsigned)ext[1] > 4096u))
            pitch = 0;          // (a handed-over box is not trusted with the products below)
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const int nbox = nrows * pitch;
        if (hg.hint && tid == 0 && half < 0 && (pitch == 0 || nbox > hg.small_cap))
            atomicAdd(hg.hint, TX / kT);   // spill feedback, in 8-wide tiles
        // self_serve: a tile that does not fit keeps an EMPTY box -- every live voxel then fails the window test
        // below and scatters its taps straight to global memory (the path of a stale handed-over box)
        bool direct_tile = false;
        if (pitch == 0 || nbox > hg.box_cap) {
            if (TX == 16 && given && half < 0 && sp.tx0 + ti * 2 + 1 < hg.tiles[2]) {
                // taken again as its two x-halves, each with the box of the forward tile it is (8 lanes of every
                // 16 sit out a pass): what the forward kernel could hold, 6144 cells can
                half_next = 0;
                continue;
            }
            if (hg.self_serve) {
                direct_tile = true;
            } else {
                const int t8 = sp.tx0 + ti * (TX / kT) + (half >= 0 ? half : tid);
                if (tid < (half >= 0 ? 1 : TX / kT) && t8 < hg.tiles[2]) {
                    const int slot = atomicAdd(&hg.spill[0], 1);
                    hg.spill[1 + slot] = sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + t8;
                }
                continue;
            }
        }
        const bool interior = b0[0] >= 0 && b0[0] + ext[0] <= hg.in_len[0] && b0[1] >= 0 &&
                              b0[1] + ext[1] <= hg.in_len[1] && b0[2] >= 0 && b0[2] + ext[2] <= hg.in_len[2];

        for (long long ss = 0; ss < hg.nsteps; ++ss, ++phase) {
            long long vol_off = 0, img_off = 0;
            if (hg.nstep)
                hot_step_offsets(hp, ss, vol_off, img_off);
            float* dst = dx + vol_off;
            float* gsum = reinterpret_cast<float*>(smem + kOffSum) + (phase & 1) * 4;
            if (ss > 0) {
                // later steps (channels) of the same tile: their own sum, after the previous flush
                float gm = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const bool inb = vy && vx && oz0 + ZSTEP * i < hg.out_len[0];
                    gval[i] = inb ? load_dy(dy, img_off + ooff0 + i * ostep, io16) : 0.f;
                    gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
                }
                gm = wave_sum(gm);
                if (lane == 0)
                    gsum[wave] = gm;
                lds_barrier();           // sum known; the previous step's flush is done with the box
            }
            const float gtot = unif((gsum[0] + gsum[1]) + (gsum[2] + gsum[3]));
            // |sum in a cell| <= max tap weight * sum over the tile of |dY|: this scale cannot overflow
            // (the 0.1 % margin covers the two roundings of the reciprocal and the product)
            constexpr float kC = (float)((2147483648.0 - 1024.0) /
                                         ((ORDER == 1 ? 1.0 : ORDER == 2 ? 0.4219 : ORDER == 3 ? 0.2963
                                           : ORDER == 4 ? 0.2150 : 0.1664) * 1.001));
            const float scale = gtot > 0.f ? fminf(kC * __frcp_rn(gtot), 3.0e38f) : 0.f;
            const float inv_scale = gtot > 0.f ? __frcp_rn(scale) : 0.f;

ED_UNROLL(ED_K2_U2)
            for (int i = 0; i < NV; ++i) {
                float gv = gval[0];
#pragma unroll
                for (int k = 1; k < NV; ++k)
                    gv = i == k ? gval[k] : gv;
                if (gv == 0.f || ED_DBG(hg.dbg, 128))
                    continue;
                int st[3];
                float fr[3];
                if (!voxel(i, st, fr))
                    continue;
                float w0[NT], w1[NT], w2[NT];
                weights_from_frac<float, ORDER>(fr[0], w0);
                weights_from_frac<float, ORDER>(fr[1], w1);
                weights_from_frac<float, ORDER>(fr[2], w2);
                const int rz = st[0] - b0[0], ry = st[1] - b0[1], rx = st[2] - b0[2];
                // boxes handed over by the forward call are a hint: a window outside goes the direct way
                const bool outside = direct_tile || (given && (rz < 0 || rz + ORDER >= ext[0] || ry < 0 || ry + ORDER >= ext[1] ||
                                                               rx < 0 || rx + ORDER >= ext[2]));
                // (A contribution is resolved to wmax * sum|dY| / 2^31 of its TILE.  Sending voxels far
                // below the tile's scale down the float path as well was measured and dropped: with a
                // threshold of 2^-20 of the tile's sum 0.05 % of the voxels of a uniform-random dY take
                // it, but 3 % of the WAVES then run the 64-iteration loop -- K2 310 -> 387 us.  The bound
                // is documented in include/edhip.h and pinned by a test instead.)
                if ((__float_as_int(gv) & 0x7f800000) == 0x7f800000 || outside) {
                    // inf / NaN gradient (no fixed-point scale), or a window outside a stale box: this
                    // voxel scatters its taps with float atomics straight to global memory (rare,
                    // rolled loop)
#pragma unroll 1
                    for (int t = 0; t < NT * NT * NT; ++t) {
                        const int l0 = t / (NT * NT), l1 = (t / NT) % NT, l2 = t % NT;
                        const int zs = mirror_i32(st[0] + l0, hg.in_len[0]);
                        const int ys = mirror_i32(st[1] + l1, hg.in_len[1]);
                        const int xs = mirror_i32(st[2] + l2, hg.in_len[2]);
                        float wp = w0[0], wq = w1[0], wr = w2[0];
#pragma unroll
                        for (int l = 1; l < NT; ++l) {
                            wp = l0 == l ? w0[l] : wp;
                            wq = l1 == l ? w1[l] : wq;
                            wr = l2 == l ? w2[l] : wr;
                        }
                        unsafeAtomicAdd(dst + (zs * hg.vol_sz + ys * hg.vol_sy + xs), gv * wp * wq * wr);
                    }
                    continue;
                }
                int* bp = box + (rz * by + ry) * pitch + rx;
                const float gs = gv * scale;
#pragma unroll
                for (int l0 = 0; l0 < NT; ++l0) {
                    const float g0 = gs * w0[l0];
#pragma unroll
                    for (int l1 = 0; l1 < NT; ++l1) {
                        const float g1 = g0 * w1[l1];
                        int* rp = bp + (l0 * by + l1) * pitch;
                        // (the NT products of a row as packed multiplies: v_pk_mul_f32 does two per issue)
                        typedef float f2_t __attribute__((ext_vector_type(2)));
                        float pr[NT + 1];
#pragma unroll
                        for (int l2 = 0; l2 + 1 < NT + 1; l2 += 2) {
                            const f2_t wv = {w2[l2], l2 + 1 < NT ? w2[l2 + 1] : 0.f};
                            const f2_t gg = {g1, g1};
                            const f2_t pv = wv * gg;
                            pr[l2] = pv.x;
                            pr[l2 + 1] = pv.y;
                        }
#pragma unroll
                        for (int l2 = 0; l2 < NT; ++l2)
                            atomicAdd(reinterpret_cast<unsigned*>(rp + l2), (unsigned)round_half_up_i32(pr[l2]));
                    }
                }
            }
            lds_barrier();               // B3: all contributions are in
            // flush: half a wave per box row, lanes along x -- one float atomic per touched source
            // element, runs of consecutive addresses (deform.c:791-813: mirror-mapped at the edges).
            // (The cells in row-major order over all lanes -- every lane live, a third fewer LDS
            // operations and instructions -- was measured: the part without the atomics drops from 25
            // to 16 us, the atomics rise from 40 to 63 us: a wave-instruction then spans three to four
            // row segments instead of two.)
            {
                constexpr int FL = TX == 8 ? 16 : 32;          // lanes per box row
                constexpr int FR = kBlock / FL;                // rows per pass
                constexpr int FU = 4;                          // rows in flight per lane: the LDS reads of
                const int sub = tid & (FL - 1);                // FU rows are issued before the first atomic
                const int rslot = tid / FL;
                const float inv_by = 1.f / (float)by;
                const int nr = (ED_DBG(hg.dbg, 64) || direct_tile) ? 0 : nrows;
                for (int xo = 0; xo < ext[2]; xo += FL) {
                    const int xi = xo + sub;
                    const bool xin = xi < ext[2];
                    const int xs = interior ? xi : mirror_i32(b0[2] + xi, hg.in_len[2]);
                    for (int r0 = rslot; r0 < nr; r0 += FU * FR) {
                        int acc[FU];
#pragma unroll
                        for (int k = 0; k < FU; ++k) {
                            const int r = r0 + k * FR;
                            // read and reset in one LDS operation (ds_wrxchg_rtn_b32)
                            acc[k] = (xin && r < nr) ? __hip_atomic_exchange(&box[r * pitch + xi], 0, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_WORKGROUP)
                                                     : 0;
                        }
#pragma unroll
                        for (int k = 0; k < FU; ++k) {
                            if (ED_DBG(hg.dbg, 4096) ? acc[k] == 0x7ffffff1 : acc[k] != 0) {     // (4096: timing without the atomics)
                                const int r = r0 + k * FR;
                                const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                                int rowoff;
                                if (interior)
                                    rowoff = (b0[0] + zr) * hg.vol_sz + (b0[1] + yr) * hg.vol_sy + b0[2];
                                else
                                    rowoff = mirror_i32(b0[0] + zr, hg.in_len[0]) * hg.vol_sz +
                                             mirror_i32(b0[1] + yr, hg.in_len[1]) * hg.vol_sy;
                                unsafeAtomicAdd(dst + (rowoff + xs), (float)acc[k] * inv_scale);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ================================================================================================
// K2 from records (hot_grad2_kernel): the gradient kernel of a step whose forward call (or a records-only
// launch in front of it) left every voxel's window start and fractions in HBM (HotGeom::rec) and every
// 8^3 tile's box (HotGeom::boxes).  No tables, no fp64, no boundary map: a voxel is 16 bytes of record +
// 4 bytes of dY.  What the kernel is bound by is the LDS atomic unit (64 ds_add_u32 per voxel in
// hot_grad_kernel, 73 % busy, profiles/r03_pmc_summary.txt), so the scatter works on PAIRS of x-neighbours:
// when the second voxel's window starts one cell after the first's in the same (z, y) rows -- 78 % of the
// pairs of the benchmark field, tools/sim/runs.py -- their 2 x 4 contributions to a row are merged in
// registers into 5 cells: 80 atomics per pair instead of 128.  SIMD form: the lanes of a wave must all be
// on the same path, so the tile's pairs are first classified from their packed starts (two words per pair)
// and compacted into two LDS work lists -- regular pairs / single voxels -- from which every wave-step is
// filled with items of one kind.  Tiles of 8 x 8 x TX voxels as before; the fixed-point scale, the
// exchange flush and the spill list are hot_grad_kernel's.
// ================================================================================================
constexpr int kG2Sum = 416;                    // float[2][4]: per-wave sums of |dY|, two tiles / steps in rotation
constexpr int kG2Cnt = 448;                    // int[3][2]: items on the two lists, three tiles in rotation
constexpr int kG2List = 512;                   // two tiles in rotation: u16[8 * 8 * TX / 2] regular pairs | u16[8 * 8 * TX] single voxels
template <int TX> constexpr int g2_list_bytes() { return 2 * (8 * 8 * TX / 2) + 2 * (8 * 8 * TX); }
template <int TX> constexpr int g2_cells() { return (kG2List + 2 * g2_list_bytes<TX>() + 15) & ~15; }

// Software pipeline over the tiles of a strip (measured on the first, unpipelined form: producer + barrier
// alone 121 of 405 us, flush 127 us against hot_grad_kernel's 40 -- on this target stores and atomics count
// in vmcnt IN ORDER with loads, so every load issued behind a flush waits for the flush's global atomics):
//   * the packed starts, dY and boxes a producer needs are requested two tiles ahead of their flush;
//   * the producer of tile t + 1 runs in front of the consumers of tile t (two sets of lists);
//   * after the barrier that closes the scatter of tile t every lane requests its first work item of tile
//     t + 1, and only then issues the flush's atomics; later items are requested one item ahead.
template <int ORDER, int TX, int WGS = 4>
__global__ __launch_bounds__(kBlock, WGS) void hot_grad2_kernel(const HotGeom hg)
{
    constexpr int NT = ORDER + 1;
    constexpr int NPX = TX / 2;                      // pairs along x
    constexpr int NI = 8 * 8 * NPX / kBlock;         // pairs per lane and tile (TX 16: 2)
    constexpr int ZSTEP = 8 / NI;
    constexpr int NK = TX / kT;                      // forward tiles per tile
    static_assert(NK <= 2, "two forward tiles per gradient tile at most");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    HotStrip sp;
    if (!hot_strip(hg, sp, blockIdx.x))
        return;
    // The thread id goes through an empty asm statement wherever per-lane constants are derived from it: the
    // compiler otherwise hoists every such value (pair coordinates, bounds tests, 64-bit addresses) out of the
    // tile loop and keeps ~100 registers live across the consumers -- 158 VGPRs, or 15 spilled at 128.
    auto fresh_tid = [] { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tb = kBlock - 1 - tid;       // single voxels are dealt to the lanes in reverse: the last wave, which gets the fewest pairs, first
    int* box = reinterpret_cast<int*>(smem + g2_cells<TX>());
    int* cnt = reinterpret_cast<int*>(smem + kG2Cnt);
    HotParams* hpw = reinterpret_cast<HotParams*>(smem);
    // the accumulator cells start at zero and every flush leaves the cells it read at zero again
    for (int e = tid * 4; e < hg.box_cap; e += kBlock * 4)
        *reinterpret_cast<int4*>(box + e) = make_int4(0, 0, 0, 0);
    if (tid < 8) {
        hpw->step_len[tid] = hg.step_len[tid];
        hpw->in_step_stride[tid] = hg.vol_step[tid];
        hpw->out_step_stride[tid] = hg.img_step[tid];
        if (tid < 6)
            cnt[tid] = 0;
        if (tid == 0)
            hpw->nstep = hg.nstep;
    }
    const HotParams* hp = hpw;

    // producer map, lane -> pair: 8 pairs along x, the 16 lanes that go through the LDS atomic unit
    // together hold rows y and y + 2 (hot_grad_kernel's map; list order follows lane order)
    auto pair_xp = [](int t) { return TX == 16 ? (t & 7) : (t & (NPX - 1)); };
    auto pair_yy = [](int t) { return TX == 16 ? 2 * ((t >> 3) & 1) + ((t >> 4) & 1) + 4 * ((t >> 5) & 1) : ((t / NPX) & 7); };
    auto pair_zq = [](int t) { return t / (NPX * 8); };
    const int ntile = (sp.ntile * kT + TX - 1) / TX;
    float* dx = hg.vol_w + sp.sample * hg.vol_bstride;
    const float* __restrict__ dy = hg.img_r + sp.sample * hg.img_bstride;
    const float4* __restrict__ rec = hg.rec + sp.sample * hg.rec_bstride;
    const int O1 = hg.out_len[1], O2 = hg.out_len[2];

    // ---- requests: what the producer of tile t needs, into registers ---------------------------------
    unsigned nw[NI][2];
    float ng[NI][2];
    int nbx = 0;              // lane 8 k + h: word h of forward tile k's box
    auto request = [&](int t) {
        const int ft = fresh_tid();
        const int xp = pair_xp(ft), oy = sp.ty * kT + pair_yy(ft), oz0 = sp.tz * kT + pair_zq(ft);
        const int ox = sp.tx0 * kT + t * TX + 2 * xp;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int oz = oz0 + ZSTEP * i;
            const bool in0 = oz < hg.out_len[0] && oy < O1 && ox < O2;
            const bool in1 = in0 && ox + 1 < O2;
            const long long ridx = ((long long)oz * O1 + oy) * O2 + ox;
            const int didx = oz * hg.img_sz + oy * hg.img_sy + ox;
            nw[i][0] = in0 ? __float_as_uint(rec[ridx].w) : kRecDead;
            ng[i][0] = in0 ? dy[didx] : 0.f;
            nw[i][1] = in1 ? __float_as_uint(rec[ridx + 1].w) : kRecDead;
            ng[i][1] = in1 ? dy[didx + 1] : 0.f;
        }
        const int t0 = sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 + t * NK;
        const int fl = ft & 63;
        const bool have = fl < 8 * NK && sp.tx0 + t * NK + (fl >> 3) < hg.tiles[2];
        nbx = have ? hg.boxes[(size_t)t0 * 8 + fl] : ((fl & 7) < 3 ? 0x7fffffff : (int)0x80000000);
    };
    // ---- producer of tile t: classify this lane's pairs from the packed starts, compact, sum |dY| -------
    auto produce = [&](int t) {
        unsigned short* la = reinterpret_cast<unsigned short*>(smem + kG2List + (t & 1) * g2_list_bytes<TX>());
        unsigned short* lb = la + 8 * 8 * NPX;
        int* tcnt = cnt + (t % 3) * 2;
        float gm = 0.f;
        bool reg[NI], live[NI][2];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned w0 = nw[i][0], w1 = nw[i][1];
            live[i][0] = !(w0 & kRecDead);
            live[i][1] = !(w1 & kRecDead);
            // one cell further along x, same rows: both voxels sit in the same 8^3 tile, so the packed
            // starts refer to one box and differ by exactly the x unit
            reg[i] = live[i][0] && live[i][1] && (w1 - w0) == 0x10000u;
            gm += (__float_as_int(ng[i][0]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(ng[i][0]);
            gm += (__float_as_int(ng[i][1]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(ng[i][1]);
        }
        gm = wave_sum(gm);
        if (lane == 0)
            reinterpret_cast<float*>(smem + kG2Sum)[(t & 1) * 4 + wave] = gm;
        // compaction: one returning LDS atomic per wave and list, ranks from the ballots
        unsigned long long ma[NI], mb[NI][2];
        int na = 0, nb = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            ma[i] = __ballot(reg[i]);
            mb[i][0] = __ballot(live[i][0] && !reg[i]);
            mb[i][1] = __ballot(live[i][1] && !reg[i]);
            na += __popcll(ma[i]);
            nb += __popcll(mb[i][0]) + __popcll(mb[i][1]);
        }
        int base_a = 0, base_b = 0;
        if (lane == 0) {
            base_a = na ? atomicAdd(&tcnt[0], na) : 0;
            base_b = nb ? atomicAdd(&tcnt[1], nb) : 0;
        }
        base_a = uni(base_a);
        base_b = uni(base_b);
        const int ft = fresh_tid();
        const unsigned long long below = (1ull << (ft & 63)) - 1ull;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pid = ((pair_zq(ft) + ZSTEP * i) * 8 + pair_yy(ft)) * NPX + pair_xp(ft);
            if (reg[i])
                la[base_a + __popcll(ma[i] & below)] = (unsigned short)pid;
            base_a += __popcll(ma[i]);
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (live[i][v] && !reg[i])
                    lb[base_b + __popcll(mb[i][v] & below)] = (unsigned short)(pid * 2 + v);
                base_b += __popcll(mb[i][v]);
            }
        }
    };
    // ---- work items -----------------------------------------------------------------------------------
    // (one register set for both kinds: a single voxel uses r0, g0, k)
    struct Item { float4 r0, r1; float g0, g1; int k; };
    auto list_a = [&](int t) { return reinterpret_cast<const unsigned short*>(smem + kG2List + (t & 1) * g2_list_bytes<TX>()); };
    auto list_b = [&](int t) { return list_a(t) + 8 * 8 * NPX; };
    auto fetch_a_id = [&](int t, const float* dys, int pid, Item& it) {
        const int pxp = pid % NPX, pyy = (pid / NPX) & 7, pz = pid / (NPX * 8);
        const int oz = sp.tz * kT + pz, oyy = sp.ty * kT + pyy, ox = sp.tx0 * kT + t * TX + 2 * pxp;
        const long long ridx = ((long long)oz * O1 + oyy) * O2 + ox;
        const int didx = oz * hg.img_sz + oyy * hg.img_sy + ox;
        it.r0 = rec[ridx];
        it.r1 = rec[ridx + 1];
        it.g0 = dys[didx];
        it.g1 = dys[didx + 1];
        it.k = (2 * pxp) / kT;
    };
    auto fetch_a = [&](int t, const float* dys, int j, Item& it) { fetch_a_id(t, dys, list_a(t)[j], it); };
    auto fetch_b_id = [&](int t, const float* dys, int vid, Item& it) {
        const int pid = vid >> 1, v = vid & 1;
        const int pxp = pid % NPX, pyy = (pid / NPX) & 7, pz = pid / (NPX * 8);
        const int oz = sp.tz * kT + pz, oyy = sp.ty * kT + pyy, ox = sp.tx0 * kT + t * TX + 2 * pxp + v;
        it.r0 = rec[((long long)oz * O1 + oyy) * O2 + ox];
        it.g0 = dys[oz * hg.img_sz + oyy * hg.img_sy + ox];
        it.k = (2 * pxp) / kT;
    };
    auto fetch_b = [&](int t, const float* dys, int j, Item& it) { fetch_b_id(t, dys, list_b(t)[j], it); };

#ifdef EDHIP_EXPERIMENTS
    // per-wave phase clocks (s_memtime ticks summed over the strip): produce, consume, wait B3, fetch + flush, wait B1
    long long tacc[5] = {0, 0, 0, 0, 0};
    long long tmark = 0;
#define ED_TICK(K) do { if (hg.dbgbuf) { const long long now_ = __builtin_readcyclecounter(); tacc[K] += now_ - tmark; tmark = now_; } } while (0)
#else
#define ED_TICK(K) do { } while (0)
#endif
    request(0);
    __syncthreads();          // cells, parameters, counters
    produce(0);
    int nbx_cur = nbx;
    if (ntile > 1)
        request(1);
    lds_barrier();            // lists of tile 0
    int n_a = uni(cnt[0]), n_b = uni(cnt[1]);
    float gtot0;        // sum of |dY| over the current tile (first step)
    {
        const float* gsum = reinterpret_cast<const float*>(smem + kG2Sum);
        gtot0 = unif((gsum[0] + gsum[1]) + (gsum[2] + gsum[3]));
    }
    Item cur;
    if (tid < n_a)
        fetch_a(0, dy, tid, cur);
    else if (tb < n_b)
        fetch_b(0, dy, tb, cur);
    int phase = 0;

    for (int ti = 0; ti < ntile; ++ti) {
        const int ox0 = sp.tx0 * kT + ti * TX;
        // ---- this tile's box: the union of the boxes of the forward tiles it covers -----------------------
        int b0[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, bhi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
        int tb0[NK][3];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                tb0[k][h] = __builtin_amdgcn_readlane(nbx_cur, 8 * k + h);
                b0[h] = min(b0[h], tb0[k][h]);
                bhi[h] = max(bhi[h], __builtin_amdgcn_readlane(nbx_cur, 8 * k + 3 + h));
            }
        }
        const bool any = bhi[0] >= b0[0] && bhi[1] >= b0[1] && bhi[2] >= b0[2];
        const int ext[3] = {bhi[0] - b0[0] + 1, bhi[1] - b0[1] + 1, bhi[2] - b0[2] + 1};
        // 16 lanes of a row hit cells two apart; pitch 8 * odd keeps rows y and y + 2 on disjoint banks
        int pitch = ext[2] <= 8 ? 8 : (ext[2] <= 24 ? 24 : (ext[2] <= 40 ? 40 : (ext[2] <= 56 ? 56 : 0)));
        if ((unsigned)ext[0] > 255u || (unsigned)ext[1] > 255u)
            pitch = 0;
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const int nbox = nrows * pitch;
        // (flush: one reflection maps every box index into the array -- always so for boxes made of window
        // starts inside the array; anything else is left to the general kernels)
        const bool simple = b0[0] > -hg.in_len[0] && b0[0] + ext[0] < 2 * hg.in_len[0] && b0[1] > -hg.in_len[1] &&
                            b0[1] + ext[1] < 2 * hg.in_len[1] && b0[2] > -hg.in_len[2] && b0[2] + ext[2] < 2 * hg.in_len[2];
        const bool fits = pitch != 0 && nbox <= hg.box_cap && simple;
        const bool work = any && fits;
        if (any && hg.hint && tid == 0 && (pitch == 0 || nbox > hg.small_cap))
            atomicAdd(hg.hint, NK);        // spill feedback, in 8-wide tiles
        if (any && !fits && tid < NK && sp.tx0 + ti * NK + tid < hg.tiles[2]) {
            const int slot = atomicAdd(&hg.spill[0], 1);
            hg.spill[1 + slot] = sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 + ti * NK + tid;
        }
        const bool interior = b0[0] >= 0 && b0[0] + ext[0] <= hg.in_len[0] && b0[1] >= 0 &&
                              b0[1] + ext[1] <= hg.in_len[1] && b0[2] >= 0 && b0[2] + ext[2] <= hg.in_len[2];
        // packed (box of the voxel's forward tile) - (this tile's box): added to a record's packed start
        unsigned delta[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k)
            delta[k] = ((unsigned)(tb0[k][0] - b0[0]) & 255u) | (((unsigned)(tb0[k][1] - b0[1]) & 255u) << 8) |
                       (((unsigned)(tb0[k][2] - b0[2]) & 255u) << 16);

        // ---- the NEXT tile's producer, in front of this tile's consumers ------------------------------
#ifdef EDHIP_EXPERIMENTS
        if (ED_DBG_PTR(hg.dbgbuf) && ti == 0)
            tmark = __builtin_readcyclecounter();
#endif
        int nbx_next = 0;
        if (ti + 1 < ntile) {
            produce(ti + 1);
            nbx_next = nbx;
        }

        ED_TICK(0);
        for (long long ss = 0; ss < hg.nsteps; ++ss, ++phase) {
            long long vol_off = 0, img_off = 0;
            if (hg.nstep)
                hot_step_offsets(hp, ss, vol_off, img_off);
            float* dst = dx + vol_off;
            const float* __restrict__ dys = dy + img_off;
            const bool last_step = ss + 1 == hg.nsteps;
            float gtot;
            if (ss == 0) {
                gtot = gtot0;
            } else {
                // later steps (channels) of the same tile: their own sum, in the NEXT tile's slot once that
                // tile's producer is done with it... kept apart instead: slot 2 (bytes 32..47 of the counters' pad)
                float* gsum = reinterpret_cast<float*>(smem + kG2Cnt + 32);
                float gs2 = 0.f;
                const int ft = fresh_tid();
                const int oy = sp.ty * kT + pair_yy(ft), oz0 = sp.tz * kT + pair_zq(ft);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int oz = oz0 + ZSTEP * i;
                    const int ox = ox0 + 2 * pair_xp(ft);
                    const int didx = oz * hg.img_sz + oy * hg.img_sy + ox;
                    const bool in0 = oz < hg.out_len[0] && oy < O1 && ox < O2;
                    const float g0 = in0 ? dys[didx] : 0.f;
                    const float g1 = (in0 && ox + 1 < O2) ? dys[didx + 1] : 0.f;
                    gs2 += (__float_as_int(g0) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(g0);
                    gs2 += (__float_as_int(g1) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(g1);
                }
                gs2 = wave_sum(gs2);
                if (lane == 0)
                    gsum[wave] = gs2;
                lds_barrier();           // (the previous step's flush ended with a barrier: slot free)
                gtot = unif((gsum[0] + gsum[1]) + (gsum[2] + gsum[3]));
                if (tid < n_a)
                    fetch_a(ti, dys, tid, cur);
                else if (tb < n_b)
                    fetch_b(ti, dys, tb, cur);
            }
            // |sum in a cell| <= max tap weight * sum over the tile of |dY|: this scale cannot overflow
            constexpr float kC = (float)((2147483648.0 - 1024.0) /
                                         ((ORDER == 1 ? 1.0 : ORDER == 2 ? 0.4219 : ORDER == 3 ? 0.2963
                                           : ORDER == 4 ? 0.2150 : 0.1664) * 1.001));
            const float scale = gtot > 0.f ? fminf(kC * __frcp_rn(gtot), 3.0e38f) : 0.f;
            const float inv_scale = gtot > 0.f ? __frcp_rn(scale) : 0.f;

            // a voxel with an inf / NaN gradient has no fixed-point scale: its taps go straight to global
            // memory with float atomics (rare, rolled loop; deform.c:791-813 for the mirror-mapped indices)
            auto direct = [&](float gv, unsigned wrel, int k, const float* w0, const float* w1, const float* w2) {
                const int st0 = (int)(wrel & 255u) + (k ? tb0[NK - 1][0] : tb0[0][0]);
                const int st1 = (int)((wrel >> 8) & 255u) + (k ? tb0[NK - 1][1] : tb0[0][1]);
                const int st2 = (int)((wrel >> 16) & 255u) + (k ? tb0[NK - 1][2] : tb0[0][2]);
#pragma unroll 1
                for (int t = 0; t < NT * NT * NT; ++t) {
                    const int l0 = t / (NT * NT), l1 = (t / NT) % NT, l2 = t % NT;
                    const int zs = mirror_i32(st0 + l0, hg.in_len[0]);
                    const int ys = mirror_i32(st1 + l1, hg.in_len[1]);
                    const int xs = mirror_i32(st2 + l2, hg.in_len[2]);
                    float wp = w0[0], wq = w1[0], wr = w2[0];
#pragma unroll
                    for (int l = 1; l < NT; ++l) {
                        wp = l0 == l ? w0[l] : wp;
                        wq = l1 == l ? w1[l] : wq;
                        wr = l2 == l ? w2[l] : wr;
                    }
                    unsafeAtomicAdd(dst + (zs * hg.vol_sz + ys * hg.vol_sy + xs), gv * wp * wq * wr);
                }
            };
#ifdef EDHIP_EXPERIMENTS
            // (profiling build, EDHIP_TILE_DBG: 4 no flush, 8 no consumers)
            const int n_a_ = (!work || ED_DBG(hg.dbg, 8)) ? 0 : n_a, n_b_ = (!work || ED_DBG(hg.dbg, 8)) ? 0 : n_b;
            const int flush_x = (!work || ED_DBG(hg.dbg, 4)) ? 0 : ext[2];
#else
            const int n_a_ = work ? n_a : 0, n_b_ = work ? n_b : 0, flush_x = work ? ext[2] : 0;
#endif
            // (the row pitch as a compile-time constant: a row's cells are immediate offsets off four plane bases)
            auto consume = [&](auto pitch_c) {
                constexpr int PITCH = decltype(pitch_c)::value;
                const int plane = by * PITCH;
                // ---- regular pairs: 5 cells per row --------------------------------------------------
                for (int j = tid; j < n_a_; j += kBlock) {
                    const Item it = cur;
                    if (j + kBlock < n_a_)
                        fetch_a(ti, dys, j + kBlock, cur);
                    else if (tb < n_b_)
                        fetch_b(ti, dys, tb, cur);       // the lane's last pair: its first single voxel next
                    const float4 r0 = it.r0, r1 = it.r1;
                    const float g0v = it.g0, g1v = it.g1;
                    const int k = it.k;
                    float wz0[NT], wy0[NT], wx0[NT], wz1[NT], wy1[NT], wx1[NT];
                    weights_from_frac<float, ORDER>(r0.x, wz0);
                    weights_from_frac<float, ORDER>(r0.y, wy0);
                    weights_from_frac<float, ORDER>(r0.z, wx0);
                    weights_from_frac<float, ORDER>(r1.x, wz1);
                    weights_from_frac<float, ORDER>(r1.y, wy1);
                    weights_from_frac<float, ORDER>(r1.z, wx1);
                    const unsigned w0 = __float_as_uint(r0.w);
                    const bool nf0 = (__float_as_int(g0v) & 0x7f800000) == 0x7f800000;
                    const bool nf1 = (__float_as_int(g1v) & 0x7f800000) == 0x7f800000;
                    if (nf0 || nf1) {
                        if (g0v != 0.f)
                            direct(g0v, w0, k, wz0, wy0, wx0);
                        if (g1v != 0.f)
                            direct(g1v, __float_as_uint(r1.w), k, wz1, wy1, wx1);
                        continue;
                    }
                    const unsigned rel = (w0 & 0xffffffu) + (k ? delta[NK - 1] : delta[0]);
                    int* bp = box + (((int)(rel & 255u) * by + (int)((rel >> 8) & 255u)) * PITCH + (int)(rel >> 16));
                    const float gs0 = g0v * scale, gs1 = g1v * scale;
#pragma unroll
                    for (int l0 = 0; l0 < NT; ++l0) {
                        const float a0 = gs0 * wz0[l0], a1 = gs1 * wz1[l0];
                        int* pl = bp + l0 * plane;
#pragma unroll
                        for (int l1 = 0; l1 < NT; ++l1) {
                            const float p0 = a0 * wy0[l1], p1 = a1 * wy1[l1];
                            int* rp = pl + l1 * PITCH;
                            float c[NT + 1];
                            c[0] = p0 * wx0[0];
#pragma unroll
                            for (int l2 = 1; l2 < NT; ++l2)
                                c[l2] = fmaf(p1, wx1[l2 - 1], p0 * wx0[l2]);
                            c[NT] = p1 * wx1[NT - 1];
#pragma unroll
                            for (int l2 = 0; l2 <= NT; ++l2)
                                atomicAdd(reinterpret_cast<unsigned*>(rp + l2), (unsigned)round_half_up_i32(c[l2]));
                        }
                    }
                }
                // ---- single voxels ---------------------------------------------------------------------
                // (a lane's first single voxel was requested with the tile's first items when the lane had no
                // pair to start with, otherwise under its last pair)
                for (int j = tb; j < n_b_; j += kBlock) {
                    const Item it = cur;
                    if (j + kBlock < n_b_)
                        fetch_b(ti, dys, j + kBlock, cur);
                    const float4 r0 = it.r0;
                    const float gv = it.g0;
                    const int k = it.k;
                    if (gv == 0.f)
                        continue;
                    float w0[NT], w1[NT], w2[NT];
                    weights_from_frac<float, ORDER>(r0.x, w0);
                    weights_from_frac<float, ORDER>(r0.y, w1);
                    weights_from_frac<float, ORDER>(r0.z, w2);
                    const unsigned wr = __float_as_uint(r0.w);
                    if ((__float_as_int(gv) & 0x7f800000) == 0x7f800000) {
                        direct(gv, wr, k, w0, w1, w2);
                        continue;
                    }
                    const unsigned rel = (wr & 0xffffffu) + (k ? delta[NK - 1] : delta[0]);
                    int* bp = box + (((int)(rel & 255u) * by + (int)((rel >> 8) & 255u)) * PITCH + (int)(rel >> 16));
                    const float gs = gv * scale;
#pragma unroll
                    for (int l0 = 0; l0 < NT; ++l0) {
                        const float a0 = gs * w0[l0];
                        int* pl = bp + l0 * plane;
#pragma unroll
                        for (int l1 = 0; l1 < NT; ++l1) {
                            const float p0 = a0 * w1[l1];
                            int* rp = pl + l1 * PITCH;
#pragma unroll
                            for (int l2 = 0; l2 < NT; ++l2)
                                atomicAdd(reinterpret_cast<unsigned*>(rp + l2), (unsigned)round_half_up_i32(p0 * w2[l2]));
                        }
                    }
                }
            };
            switch (pitch) {
            case 8: consume(std::integral_constant<int, 8>()); break;
            case 24: consume(std::integral_constant<int, 24>()); break;
            case 40: consume(std::integral_constant<int, 40>()); break;
            default: consume(std::integral_constant<int, 56>()); break;
            }
            // (requests for the producer of tile ti + 2: behind the consumers, whose registers they would
            // crowd, still in front of this tile's flush)
            if (last_step && ti + 2 < ntile)
                request(ti + 2);
            ED_TICK(1);
            lds_barrier();               // B3: all contributions are in; the next tile's lists are complete
            ED_TICK(2);
            if (tid < 2)
                cnt[(ti % 3) * 2 + tid] = 0;      // this tile's counts: read by every wave before this barrier; next used by tile ti + 3
            // Behind the barrier every LDS operation of this wave queues behind the other workgroups' scatter
            // atomics, so what the rest of the round needs from LDS is requested in ONE batch: the next tile's
            // counts and sums, this lane's first entries of its two lists (speculative: used only below the
            // counts) and the exchanges of the flush's first pass.
            // Flush: a lane owns a column of the box -- one (y, x) and every z -- so that a cell's address is the
            // previous one plus a plane (LDS) / a slice of the volume (global); the columns of the (y, x) plane are
            // dealt to the 256 lanes in row-major order (one pass for the usual ~12 x 20 plane).  The row-walking
            // form spent two divisions and two products per cell, left a third of the lanes idle and took a third
            // of the wave's time (tools/g2_phases.py: the flush is bound by its instruction count).
            constexpr int FZ = 16;
            const int ftf = fresh_tid();
            const int plane_cells = by * pitch;
            const int ncol = flush_x > 0 ? by * ext[2] : 0;
            const float inv_ex = 1.f / (float)ext[2];
            int next_a = 0, next_b = 0, first_a = 0, first_b = 0;
            float4 next_sum = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool more = last_step && ti + 1 < ntile;
            if (more) {
                next_a = cnt[((ti + 1) % 3) * 2];
                next_b = cnt[((ti + 1) % 3) * 2 + 1];
                next_sum = *reinterpret_cast<const float4*>(smem + kG2Sum + ((ti + 1) & 1) * 16);
                first_a = list_a(ti + 1)[tid];
                first_b = list_b(ti + 1)[tb];
            }
            // cells (z0 .. z0 + FZ) of column c: read and reset in one LDS operation each
            auto exchange = [&](int c, int z0, int (&acc)[FZ]) {
#pragma unroll
                for (int q = 0; q < FZ; ++q)
                    acc[q] = 0;
                if (c < ncol) {
                    const int yi = (int)(((float)c + 0.5f) * inv_ex), xi = c - yi * ext[2];
                    // (planes beyond the box: the last plane once more -- it reads the zero the first visit left)
                    int* cp = box + yi * pitch + xi;
#pragma unroll
                    for (int q = 0; q < FZ; ++q)
                        acc[q] = __hip_atomic_exchange(cp + min(z0 + q, ext[0] - 1) * plane_cells, 0, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            // one float atomic per touched source element.  A box that sticks out of the array holds the taps of
            // windows at the array's ends, which the reference mirror-maps (deform.c:791-813); the window starts
            // themselves lie inside the array, so one reflection is all a box index ever needs (`simple`, above).
            auto mirror1 = [](int i, int n) { return min(max(i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i), 0), n - 1); };
            auto emit = [&](int c, int z0, const int (&acc)[FZ]) {
                const int yi = (int)(((float)c + 0.5f) * inv_ex), xi = c - yi * ext[2];
                if (interior) {
                    float* col = dst + ((b0[0] + z0) * hg.vol_sz + (b0[1] + yi) * hg.vol_sy + b0[2] + xi);
#pragma unroll
                    for (int q = 0; q < FZ; ++q) {
                        if (acc[q] != 0)
                            unsafeAtomicAdd(col + q * hg.vol_sz, (float)acc[q] * inv_scale);
                    }
                } else {
                    float* col = dst + (mirror1(b0[1] + yi, hg.in_len[1]) * hg.vol_sy + mirror1(b0[2] + xi, hg.in_len[2]));
                    int zv = b0[0] + z0;
                    asm volatile("" : "+v"(zv));        // (vector arithmetic on purpose: scalar mirror maps of every plane cost 270 spilled SGPRs)
#pragma unroll
                    for (int q = 0; q < FZ; ++q) {
                        if (acc[q] != 0)
                            unsafeAtomicAdd(col + mirror1(zv + q, hg.in_len[0]) * hg.vol_sz, (float)acc[q] * inv_scale);
                    }
                }
            };
            int acc0[FZ];
            exchange(ftf, 0, acc0);
            if (more) {
                // the next tile's first items, requested in front of this tile's flush atomics
                next_a = uni(next_a);
                next_b = uni(next_b);
                gtot0 = unif((next_sum.x + next_sum.y) + (next_sum.z + next_sum.w));
                if (tid < next_a)
                    fetch_a_id(ti + 1, dy, first_a, cur);
                else if (tb < next_b)
                    fetch_b_id(ti + 1, dy, first_b, cur);
            }
            if (ncol > 0) {
                emit(ftf, 0, acc0);
                for (int c0 = 0; c0 < ncol; c0 += kBlock)
                    for (int z0 = c0 == 0 ? FZ : 0; z0 < ext[0]; z0 += FZ) {
                        int acc[FZ];
                        exchange(c0 + ftf, z0, acc);
                        emit(c0 + ftf, z0, acc);
                    }
            }
            ED_TICK(3);
            lds_barrier();               // B1: the cells are back at zero
            ED_TICK(4);
            if (last_step) {
                n_a = next_a;
                n_b = next_b;
            }
        }
        nbx_cur = nbx_next;
    }
#ifdef EDHIP_EXPERIMENTS
    if (ED_DBG_PTR(hg.dbgbuf) && lane == 0) {
        unsigned long long* d = hg.dbgbuf + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int q = 0; q < 5; ++q)
            d[q] = (unsigned long long)tacc[q];
        d[5] = (unsigned long long)ntile;
    }
#endif
#undef ED_TICK
}

template <int ORDER>
hipError_t launch_order(const HotGeom& hg, bool gradient, unsigned nblk, size_t lds, hipStream_t stream)
{
    if (gradient) {
#ifdef EDHIP_EXPERIMENTS
        const int duo = ed_env("EDHIP_GRAD_DUO") ? atoi(ed_env("EDHIP_GRAD_DUO")) : 0;
        if (duo && !hg.has_affine && ORDER == 3) {
            if constexpr (ORDER == 3) {
                auto kern = hot_grad_kernel<ORDER, false, 4, 16, 2>;
                static bool once = false;
                if (!once) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    once = true;
                }
                const unsigned per = nblk / 8;
                hipLaunchKernelGGL(kern, dim3(8 * ((per + 1) / 2)), dim3(2 * kBlock), 2 * (size_t)hg.lds_grp, stream, hg);
            }
            return hipGetLastError();
        }
#endif
        if (hg.io16) {
            if constexpr (ORDER <= 3) {
                if (hg.has_affine)
                    hipLaunchKernelGGL((hot_grad_kernel<ORDER, true, 4, 16, 1, true>), dim3(nblk), dim3(kBlock), lds, stream, hg);
                else
                    hipLaunchKernelGGL((hot_grad_kernel<ORDER, false, 4, 16, 1, true>), dim3(nblk), dim3(kBlock), lds, stream, hg);
                return hipGetLastError();
            }
            return hipErrorNotSupported;
        }
        if (hg.has_affine)
            hipLaunchKernelGGL((hot_grad_kernel<ORDER, true, 4, 16>), dim3(nblk), dim3(kBlock), lds, stream, hg);
        else
            hipLaunchKernelGGL((hot_grad_kernel<ORDER, false, 4, 16>), dim3(nblk), dim3(kBlock), lds, stream, hg);
    } else {
        if (hg.io16) {
            if constexpr (ORDER <= 3) {
                if (hg.has_affine)
                    hipLaunchKernelGGL((hot_fwd_kernel<ORDER, true, 0, kBlock, 4, false, true>), dim3(nblk), dim3(kBlock), lds, stream, hg);
                else
                    hipLaunchKernelGGL((hot_fwd_kernel<ORDER, false, 0, kBlock, 4, false, true>), dim3(nblk), dim3(kBlock), lds, stream, hg);
                return hipGetLastError();
            }
            return hipErrorNotSupported;
        }
#ifdef EDHIP_EXPERIMENTS
        // profiling builds of the forward kernel (see hot_fwd_kernel's ABL switches)
        if (!hg.has_affine && ORDER == 3 && ed_env("EDHIP_HOT_ABL")) {
            if constexpr (ORDER == 3) {
                switch (atoi(ed_env("EDHIP_HOT_ABL"))) {
#define ED_ABL_CASE(A) case A: hipLaunchKernelGGL((hot_fwd_kernel<ORDER, false, A>), dim3(nblk), dim3(kBlock), lds, stream, hg); break;
                ED_ABL_CASE(16) ED_ABL_CASE(1024) ED_ABL_CASE(4096) ED_ABL_CASE(5120) ED_ABL_CASE(7168) ED_ABL_CASE(3072) ED_ABL_CASE(2) ED_ABL_CASE(4) ED_ABL_CASE(6) ED_ABL_CASE(46) ED_ABL_CASE(32)
#undef ED_ABL_CASE
                default: hipLaunchKernelGGL((hot_fwd_kernel<ORDER, false>), dim3(nblk), dim3(kBlock), lds, stream, hg); break;
                }
            }
            return hipGetLastError();
        }
        if (!hg.has_affine && ed_env("EDHIP_HOT_NTH512")) {
            if (atoi(ed_env("EDHIP_HOT_NTH512")) == 8)
                hipLaunchKernelGGL((hot_fwd_kernel<ORDER, false, 0, 512, 8>), dim3(nblk), dim3(512), lds, stream, hg);
            else
                hipLaunchKernelGGL((hot_fwd_kernel<ORDER, false, 0, 512, 6>), dim3(nblk), dim3(512), lds, stream, hg);
            return hipGetLastError();
        }
#endif
        if (hg.has_affine)
            hipLaunchKernelGGL((hot_fwd_kernel<ORDER, true>), dim3(nblk), dim3(kBlock), lds, stream, hg);
        else
            hipLaunchKernelGGL((hot_fwd_kernel<ORDER, false>), dim3(nblk), dim3(kBlock), lds, stream, hg);
    }
    return hipGetLastError();
}

}  // namespace

// records-only launch of K1 (first half of a gradient call without a forward call to lean on): same grid
// and LDS as the forward launch
hipError_t launch_hot_records(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream)
{
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(kBlock), lds, stream, hg);
        return hipGetLastError();
    };
#ifndef EDHIP_EXPERIMENTS
    (void)go;
    (void)order;
    return hipErrorNotSupported;      // (the records route is measured in the profiling build only, deform_tile.hip)
#else
    switch (order * 2 + (hg.has_affine ? 1 : 0)) {
    case 2: return go(hot_fwd_kernel<1, false, 0, kBlock, 4, true>);
    case 3: return go(hot_fwd_kernel<1, true, 0, kBlock, 4, true>);
    case 4: return go(hot_fwd_kernel<2, false, 0, kBlock, 4, true>);
    case 5: return go(hot_fwd_kernel<2, true, 0, kBlock, 4, true>);
    case 6: return go(hot_fwd_kernel<3, false, 0, kBlock, 4, true>);
    case 7: return go(hot_fwd_kernel<3, true, 0, kBlock, 4, true>);
    default: return hipErrorNotSupported;
    }
#endif
}

// K2 from records: LDS = parameters | sums | counters | two work lists | cells
size_t hot_grad2_lds_bytes(int* box_cap, bool large)
{
    size_t cells = large ? 44 * 1024 : 32 * 1024;      // 3 / 4 workgroups per CU
    if (const char* kb = ed_env("EDHIP_G2_CELLS_KB"))
        cells = (size_t)atoi(kb) * 1024;
    *box_cap = (int)(cells / 4);
    return (size_t)g2_cells<16>() + cells;
}

hipError_t launch_hot_grad2(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream)
{
#ifdef EDHIP_EXPERIMENTS
    if (order == 3 && ed_env("EDHIP_G2_WG5")) {
        hipLaunchKernelGGL((hot_grad2_kernel<3, 16, 5>), dim3(nblk), dim3(kBlock), lds, stream, hg);
        return hipGetLastError();
    }
#endif
#ifndef EDHIP_EXPERIMENTS
    return hipErrorNotSupported;      // (profiling build only, like the records it reads)
#else
    switch (order) {
    case 1: hipLaunchKernelGGL((hot_grad2_kernel<1, 16>), dim3(nblk), dim3(kBlock), lds, stream, hg); break;
    case 2: hipLaunchKernelGGL((hot_grad2_kernel<2, 16>), dim3(nblk), dim3(kBlock), lds, stream, hg); break;
    case 3: hipLaunchKernelGGL((hot_grad2_kernel<3, 16>), dim3(nblk), dim3(kBlock), lds, stream, hg); break;
    default: return hipErrorNotSupported;
    }
    return hipGetLastError();
#endif
}

// LDS: x table | reduction slots | wave sums | parameters | 64 Q rows | box.  Returns 0 when the
// control grid is too wide for a useful box (the general kernels take the call).
size_t hot_lds_bytes(bool gradient, int ncpx, int* box_cap, int* off_box, bool large)
{
    const size_t q = (size_t)kT * kT * 32 * (size_t)ncpx;
    const size_t off = (kOffQ + q + 15) & ~(size_t)15;
    *off_box = (int)off;
    // Large boxes (three workgroups per CU instead of four) for strongly deformed volumes.  256^3 order 3,
    // whole call, standard -> large: sigma 5 forward 262 -> 286 us, gradient 325 -> 357; sigma 10 forward
    // 368 -> 340, gradient 501 -> 427; sigma 15 forward 614 -> 496, gradient 1412 -> 1264
    // (profiles/r03_bench_misc.txt): launch_tile picks them when recent calls of the geometry spilled.
    if (gradient) {
        size_t box = large ? 36 * 1024 : kGradBoxBytes;
        if (const char* kb = ed_env("EDHIP_GRAD_BOX_KB"))
            box = (size_t)atoi(kb) * 1024;
        *box_cap = (int)(box / 4);
        const size_t total = off + box;
        return total <= 64 * 1024 ? total : 0;
    }
    // forward: two shifted float copies; 4 workgroups per CU -> 40960 bytes each (wide control
    // grids: a 64 KiB block, fewer workgroups per CU)
    size_t budget = large ? 52 * 1024 : 40 * 1024;
    if (const char* kb = ed_env("EDHIP_HOT_FWD_KB"))      // experiment: fewer workgroups per CU
        budget = (size_t)atoi(kb) * 1024;
    if (const char* abl = ed_env("EDHIP_HOT_ABL")) {      // experiments (see hot_fwd_kernel)
        const int a = atoi(abl);
        if (a & 1024) {                                    // Q rows stay in global memory
            *off_box = kOffQ;
            size_t cap = (((a & 2048) ? 32 : 40) * 1024 - kOffQ) / 8;
            cap = ((cap - 56) / 64) * 64 + 56;
            *box_cap = (int)cap;
            return kOffQ + 2 * 4 * cap;
        }
    }
    if (off + 2 * 4 * 2488 > budget)
        budget = 64 * 1024;
    if (off + 2 * 4 * 2488 > budget)
        return 0;
    size_t cap = (budget - off) / 8;
    cap = ((cap - 56) / 64) * 64 + 56;        // cap = 56 (mod 64): the copies sit on disjoint banks
    *box_cap = (int)cap;
    return off + 2 * 4 * cap;
}

hipError_t launch_hot_level1(const HotGeom& hg, int order, bool gradient, unsigned nblk, size_t lds,
                             hipStream_t stream)
{
    switch (order) {
    case 1: return launch_order<1>(hg, gradient, nblk, lds, stream);
    case 2: return launch_order<2>(hg, gradient, nblk, lds, stream);
    case 3: return launch_order<3>(hg, gradient, nblk, lds, stream);
    case 4: return launch_order<4>(hg, gradient, nblk, lds, stream);
    case 5: return launch_order<5>(hg, gradient, nblk, lds, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace tile
}  // namespace ed
