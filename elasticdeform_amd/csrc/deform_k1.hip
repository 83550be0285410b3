// deform_k1.hip -- K1, the forward gather of the benchmark case: float32 volumes, 3 deformed axes, unit stride
// along x on both sides, spline orders 1-3 (deform.c:649-924).  Round 5: rebuilt around the VALU issue budget
// (profiles/r04_pmc_summary.txt: the round-4 kernel issued 340 vector instructions per 64 voxels, 84 of them
// the gather's FMAs).  What is different from the kernel it replaces (csrc/experiments/deform_hot_r4.hip):
//
//   * SAMPLED TILE BOXES.  The source box of an 8^3 output tile used to be the exact bounding box of the tile's
//     512 tap windows: every voxel's window start went through 12 min / max, a 36-step DPP reduction and six LDS
//     atomics per wave, and the box was known only behind a barrier.  Now one wave per tile evaluates the
//     coordinate at a 4 x 4 x 4 lattice of the tile's voxels (0, 2, 4, 7 along each axis) in the strip prologue,
//     widens the range by a margin (HotParams::slack, a fraction of the rigorous interpolation bound; the tables
//     kernel measures the control grid for it) and derives everything the tile loop needs -- box, row pitch,
//     "fits", "interior" -- once, into a 64-byte record in LDS.
//   * The sampled box is an ESTIMATE, so the kernel checks it: a voxel whose window is not inside its tile's box
//     raises a flag in its lane (six integer instructions per voxel), and a wave that saw a flag redoes its
//     voxels of the strip behind the tile loop -- general coordinates, taps straight from global memory
//     (k1_fix, the path that also serves tiles whose box does not fit LDS).  Correctness never rests on the
//     margin: a margin that is too small costs time, and on the benchmark field no voxel of 16.7 M misses.
//   * FAST TILES.  A full tile whose sampled coordinate range, margin included, stays inside the array needs no
//     boundary map: its voxels skip the "left the array?" tests, the constant / valid flags and their branches
//     (a third of the old coordinate phase).  A voxel that leaves the array anyway has a window outside the
//     box and is caught by the check above.  Every other tile (array faces, partial tiles) takes the general
//     coordinates; its box is SAMPLED too (the mapped coordinate at the same 64 lattice points, plus the margin)
//     and every voxel's window is checked against it -- correctness rests on that check and on k1_fix, as for
//     the fast tiles.
//   * Q rows are staged as [control column][row][component]: the two voxels of a lane differ by an immediate
//     offset, a voxel's table reads need no address arithmetic, and the 8 rows of a wave-instruction are
//     contiguous 32-byte pieces (the old [row][column] layout spread them 160 bytes apart).
//   * Box-relative addresses: one multiply-add chain per voxel with the tile-constant part folded into a
//     per-lane offset; the shifted copy is selected arithmetically from the parity of the element offset.
//
// Unchanged: the tiling (a 256-thread workgroup per strip of <= 4 tiles along x, a lane owns a (y, x) column
// and two z slices), LDS-DMA staging of two copies of the box one element apart, the software pipeline
// (coordinates of tile t + 1 under the copies of tile t), the separable gather in x, y, z order from zero --
// the same sums, bit for bit, as every other level (crop identity full[crop] == cropped, README.md:113).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ed_device.h"
#include "ed_params.h"
#include "ed_tile.h"

namespace ed {
namespace tile {

#ifdef EDHIP_K1_STATS
// (-DEDHIP_K1_STATS, tools/k1_stats.py; the counters cost the kernel registers) [0] waves that ran k1_fix for a window outside a sampled box, [1] voxels they redid,
// [2] voxels of unfit tiles served by k1_fix, [3] fast tiles, [4] general tiles, [5] unfit tiles
__device__ unsigned long long g_k1_stats[8];
#endif

namespace {

#ifndef ED_K1_STRIP
#define ED_K1_STRIP 4
#endif
constexpr int kK1Strip = ED_K1_STRIP;                    // tiles per strip, at most
constexpr int kK1Tab = 0;                                // AxTab[kK1Strip * 8]; idx = byte offset of the control column
constexpr int kK1Rec = kK1Tab + kK1Strip * kT * 48;      // TileRec[kK1Strip]
constexpr int kK1Red = kK1Rec + kK1Strip * 64;           // (128 bytes, unused since the general tiles' boxes are sampled: the offsets behind stay put)
constexpr int kK1Hot = kK1Red + kK1Strip * 32;           // HotParams
constexpr int kK1Q = kK1Hot + 416;                       // Q[control column][64 rows][4 doubles]
constexpr int kQCol = 64 * 32;                           // bytes per control column
static_assert(kK1Q % 16 == 0, "LDS carve alignment");

// what the tile loop needs to know about a tile, derived once per strip
struct TileRec {
    int b0[3];          // box origin, in window-start (tap index) space
    int flags;
    int ext[3];         // box extents
    int pitch;          // floats per box row: 16 or 48 (0: too wide)
    int nrows;          // ext[0] * ext[1]
    int plane;          // ext[1] * pitch
    int goff;           // element offset of the box origin in the volume (meaningful for kTDma + kTZYin)
    int pad_[5];
};
static_assert(sizeof(TileRec) == 64, "TileRec layout");
enum : int {
    kTAny = 1,          // some voxel of the tile is gathered
    kTStaged = 2,       // ... and the box fits: staged, gathered from LDS
    kTFast = 4,         // full tile, coordinates inside the array: no boundary tests
    kTDma = 8,          // the box rows (and the shifted copy's extra element) lie inside the array along x: LDS-DMA
    kTXin = 16,         // the box lies inside the array along x
    kTZYin = 32,        // ... and along z and y: no mirror map of plane / row indices while staging
    kTUnfit = 64,       // the box does not fit (self-serve: k1_fix gathers the tile from global memory)
    kTGen = 128,        // general tile (array faces, partial tiles): boundary map, constant / valid flags
};

#define ED_RED6(CTRL)                                      \
    "v_min_i32_dpp %0, %0, %0 " CTRL "\n\t"                \
    "v_min_i32_dpp %1, %1, %1 " CTRL "\n\t"                \
    "v_min_i32_dpp %2, %2, %2 " CTRL "\n\t"                \
    "v_max_i32_dpp %3, %3, %3 " CTRL "\n\t"                \
    "v_max_i32_dpp %4, %4, %4 " CTRL "\n\t"                \
    "v_max_i32_dpp %5, %5, %5 " CTRL "\n\t"
// min of lo[3] / max of hi[3] over the wave's 64 lanes (all active), result in lane 63: six interleaved DPP
// chains (hand-written: the compiler turns every update_dpp step into four instructions)
__device__ __forceinline__ void wave_box63(int (&lo)[3], int (&hi)[3])
{
    asm volatile("s_nop 1\n\t"
                 ED_RED6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 ED_RED6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_half_mirror row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_mirror row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 ED_RED6("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
}
#undef ED_RED6
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// workgroup barrier that orders LDS traffic only (the output stores in flight need no draining)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// barrier that also retires this wave's LDS-DMA copies
__device__ __forceinline__ void dma_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void glds16(const float* g, float* lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <bool OUT16>
__device__ __forceinline__ void store_out(float* img, long long off, float val, int io16)
{
    if constexpr (OUT16)      // (cached 2-byte stores: streamed past the L2 their half sectors cost 17 us)
        reinterpret_cast<unsigned short*>(img)[off] = (unsigned short)narrow16(val, io16);
    else
        __builtin_nontemporal_store(val, img + off);
}

struct K1Strip {
    int tz, ty, tx0, ntile, sample;
};
// strips are dealt to the 8 XCDs in contiguous chunks (block b runs on XCD b % 8): neighbouring strips, whose
// source boxes overlap, share an L2
__device__ __forceinline__ bool k1_strip(const HotGeom& hg, K1Strip& sp, int b)
{
    const int per = (hg.total_strips + 7) >> 3;
    int s = (b & 7) * per + (b >> 3);
    if ((b >> 3) >= per || s >= hg.total_strips)
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

// ---- strip prologue: x table, Q rows, uniform parameters -> LDS; ends with a barrier -------------------------
__device__ __forceinline__ void k1_prologue(const HotGeom& hg, const K1Strip& sp, char* smem, int tid)
{
    {   // x table: 12 dwords per output column; the control-column indices (dwords 8..11, counted in doubles of a
        // Q row [ncpx][4]) become byte offsets of the column in the LDS layout
        const int* src = reinterpret_cast<const int*>(hg.xt + sp.tx0 * kT);
        int* dst = reinterpret_cast<int*>(smem + kK1Tab);
        const int avail = (hg.out_len[2] - sp.tx0 * kT) * 12;
        for (int e = tid; e < sp.ntile * kT * 12; e += kBlock) {
            const int v = e < avail ? src[e] : 0;
            dst[e] = (e % 12) >= 8 ? v * (kQCol / 4) : v;
        }
    }
    {   // Q rows (zi, yy) -> (oz, oy): 4 threads per row, 16 bytes at a time, transposed to [column][row]
        const int row16 = 2 * hg.ncpx;
        const int r = tid >> 2;
        const int oz = min(sp.tz * kT + (r >> 3), hg.out_len[0] - 1);
        const int oy = min(sp.ty * kT + (r & 7), hg.out_len[1] - 1);
        // (wide control grids: Q is laid out per x-strip, hg.ncpx columns each -- TileGeom::q_win)
        const long long qrow_id = hg.q_strips > 1
                                      ? ((long long)oz * hg.out_len[1] + oy) * hg.q_strips + sp.tx0 / hg.strip_tiles
                                      : (long long)oz * hg.out_len[1] + oy;
        const double2* src = reinterpret_cast<const double2*>(hg.q + sp.sample * hg.q_bstride + qrow_id * (4 * hg.ncpx));
        for (int p = tid & 3; p < row16; p += 4)
            *reinterpret_cast<double2*>(smem + kK1Q + (p >> 1) * kQCol + r * 32 + (p & 1) * 16) = src[p];
    }
    if (tid >= 128 && tid < 128 + 12) {
        HotParams* hp = reinterpret_cast<HotParams*>(smem + kK1Hot);
        const int k = tid - 128;
        hp->affine[k] = hg.affine[k];
        if (k < 3) {
            hp->offd[k] = (double)hg.off[k];
            hp->last[k] = (double)(hg.in_len[k] - 1);
            hp->period[k] = hg.period[k];
            hp->inv_period[k] = hg.inv_period[k];
            hp->slack[k] = hg.slack[sp.sample * 4 + k];
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

__device__ __forceinline__ void k1_step_offsets(const HotParams* hp, long long ss, long long& vol_off, long long& img_off)
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

// x table entry of one output column of the strip: cubic weights + LDS addresses of the lane's Q row in the four
// control columns (`row` = byte offset of the row inside a column)
struct XEnt {
    double w[4];
    int qa[4];
};
__device__ __forceinline__ void k1_xent(const char* smem, int col, int row, XEnt& xe)
{
    const AxTab& t = reinterpret_cast<const AxTab*>(smem + kK1Tab)[col];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        xe.w[l] = t.w[l];
        xe.qa[l] = t.idx[l] + (kK1Q + row);
    }
}
// displacement of one voxel (deform.c:693-758 after the contraction over z and y): 4 control columns x 3
// components, ds_read_b128 (components 0, 1) + ds_read_b64 (2) per column; ROFF = immediate row offset
template <int ROFF>
__device__ __forceinline__ void k1_disp(const char* smem, const XEnt& xe, double (&d)[3])
{
    double2 q01[4];
    double q2[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        q01[l] = *reinterpret_cast<const double2*>(smem + xe.qa[l] + ROFF);
        q2[l] = *reinterpret_cast<const double*>(smem + xe.qa[l] + ROFF + 16);
    }
    d[0] = xe.w[0] * q01[0].x;
    d[1] = xe.w[0] * q01[0].y;
    d[2] = xe.w[0] * q2[0];
#pragma unroll
    for (int l = 1; l < 4; ++l) {
        d[0] = fma(xe.w[l], q01[l].x, d[0]);
        d[1] = fma(xe.w[l], q01[l].y, d[1]);
        d[2] = fma(xe.w[l], q2[l], d[2]);
    }
}

// general coordinates of one voxel (deform.c:771-824): window start and fractions with the boundary map for the
// axes along which the source point left the array -- coord_axis_fast / coord_axis_mapped of ed_tile.h, the
// arithmetic every tile kernel shares.  `b[h]` = output index + crop offset (no affine) or 0 (affine: the real
// base is in `P`).  Returns true when the voxel maps to the constant.
template <int ORDER, bool AFFINE>
__device__ __forceinline__ bool k1_coords(const HotGeom& hg, const HotParams* hp, const double (&d)[3], const int (&b)[3],
                                          const double (&P)[3], int* start, float* frac, int* raw_start = nullptr)
{
    int ci[3];
    bool inr[3];
#pragma unroll
    for (int h = 0; h < 3; ++h)
        inr[h] = coord_axis_fast<ORDER, float>(AFFINE ? P[h] + d[h] : d[h], AFFINE ? 0 : b[h], hg.in_len[h], ci[h], frac[h]);
    if (raw_start) {         // the window start a fast tile's voxel works with: no range test, no boundary map
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
                cst = coord_axis_mapped<ORDER, float>(AFFINE ? P[h] + d[h] : (double)b[h] + d[h], hg.in_len[h], hg.mode,
                                                      hp->period[h], hp->inv_period[h], ci[h], frac[h]) || cst;
        }
    }
#pragma unroll
    for (int h = 0; h < 3; ++h)
        start[h] = cst ? 0 : ci[h] - ORDER / 2;
    return cst;
}

// ---- tile records ------------------------------------------------------------------------------------------------
// One lane per tile: everything the tile loop needs, from the box [lo, hi] in window-start space (hi: last tap,
// along x the padding tap of even orders included), plus the tile's side effects -- spill feedback, spill list,
// the box for the gradient call (EDHIP_FLAG_KEEP_BOXES).
__device__ __forceinline__ void k1_derive(const HotGeom& hg, const K1Strip& sp, TileRec* rec, int t, const int (&lo)[3],
                                          const int (&hi)[3], int kpad, bool fast)
{
    const bool any = hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2];
    unsigned ext[3];
#pragma unroll
    for (int h = 0; h < 3; ++h)
        ext[h] = (unsigned)hi[h] - (unsigned)lo[h] + 1u;
    const int pitch = ext[2] <= 16u ? 16 : (ext[2] <= 48u ? 48 : 0);
    const bool sane = any && ext[0] <= 1024u && ext[1] <= 1024u;
    const int nrows = sane ? (int)(ext[0] * ext[1]) : 0;
    const bool fits = sane && pitch > 0 && nrows * pitch <= hg.box_cap;
    // rows (with the shifted copy's extra element) inside the array along x: LDS-DMA, whole rows; planes / rows
    // beyond the array's z / y ends are mirror-mapped as planes / rows
    const bool dma = fits && lo[2] >= 0 && lo[2] + pitch + 1 <= hg.in_len[2];
    const bool xin = fits && lo[2] >= 0 && lo[2] + (int)ext[2] <= hg.in_len[2];
    const bool zyin = fits && lo[0] >= 0 && lo[0] + (int)ext[0] <= hg.in_len[0] && lo[1] >= 0 &&
                      lo[1] + (int)ext[1] <= hg.in_len[1];
    TileRec r;
    r.flags = (any ? kTAny : 0) | (fits ? kTStaged : 0) | (fast && fits ? kTFast : 0) | (dma ? kTDma : 0) |
              (xin ? kTXin : 0) | (zyin ? kTZYin : 0) | (any && !fits ? kTUnfit : 0) | (fast ? 0 : kTGen);
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        r.b0[h] = lo[h];
        r.ext[h] = (int)ext[h];
    }
    r.pitch = pitch;
    r.nrows = nrows;
    r.plane = (int)ext[1] * pitch;
    r.goff = (dma && zyin) ? lo[0] * hg.vol_sz + lo[1] * hg.vol_sy + lo[2] : 0;
    int4* dst = reinterpret_cast<int4*>(rec + t);
    dst[0] = make_int4(r.b0[0], r.b0[1], r.b0[2], r.flags);
    dst[1] = make_int4(r.ext[0], r.ext[1], r.ext[2], r.pitch);
    dst[2] = make_int4(r.nrows, r.plane, r.goff, 0);
#ifdef EDHIP_K1_STATS
    atomicAdd(&g_k1_stats[(r.flags & kTUnfit) ? 5 : ((r.flags & kTFast) ? 3 : 4)], 1ull);
#endif
    const int tile_id = sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 + t;
    if (any && hg.hint && !(sane && pitch > 0 && nrows * pitch <= hg.small_cap))
        atomicAdd(hg.hint, 1);             // spill feedback: would not fit the standard box
    if (any && !fits && !hg.self_serve) {  // hand the whole tile to the general kernels
        const int slot = atomicAdd(&hg.spill[0], 1);
        hg.spill[1 + slot] = tile_id;
    }
    if (hg.boxes) {
        int* bx = hg.boxes + (size_t)tile_id * 8;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            bx[h] = lo[h];
            bx[3 + h] = hi[h] - ((h == 2 && any) ? kpad : 0);      // (without the forward gather's padding tap)
        }
    }
}

// 64-tap (order 3) separable gather of one voxel from the staged box; PITCH is a template argument so that the
// row offsets are immediates.  `bp` points at tap (0, 0, 0) in the copy whose shift matches the parity of the
// window's x start: every x-run is a sequence of aligned 8-byte reads.  The reads are kept apart (an empty asm
// statement between them): the backend otherwise fuses neighbours into ds_read2_b64, which moves 16 bytes per
// lane in 16 LDS cycles where two ds_read_b64 take 2 x 2.8 (profiles/r02_ubench_lds.txt).
template <int ORDER, int PITCH, int SPLIT>
__device__ __forceinline__ float k1_gather(const float* bp, int plane, const float* w0, const float* w1, const float* w2)
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
                if (SPLIT)
                    ED_NO_DS_MERGE();
                a2 = fmaf(w2[l2], pr.x, a2);
                a2 = fmaf(w2[l2 + 1], pr.y, a2);
            }
            a1 = fmaf(w1[l1], a2, a1);
        }
        a0 = fmaf(w0[l0], a1, a0);
    }
    return a0;
}

// Voxels the tile loop did not (or may not have) served, straight from global memory, behind the loop: every
// voxel of a tile whose box does not fit LDS (self-serve), and the voxels of a fast tile whose window is not
// inside the tile's sampled box (this wave raised its flag).  General coordinates from the strip's tables in LDS,
// taps mirror-mapped per axis (deform.c:791-813), accumulation x, y, z as chains of fused multiply-adds from
// zero: the bits every other level gives.  Per wave, no barriers.  (Inlined on purpose: as a real call the kernel's
// argument block is copied to scratch for it, and every pointer loaded back from there is a flat pointer.)
template <int ORDER, bool AFFINE, bool OUT16>
__device__ __forceinline__ void k1_fix(const HotGeom& hg, const K1Strip& sp, char* smem, bool missed, int io16)
{
    constexpr int NT = ORDER + 1;
    constexpr int kPadX = NT & 1;
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem + kK1Hot);
    const TileRec* rec = reinterpret_cast<const TileRec*>(smem + kK1Rec);
    const int tid = threadIdx.x;
    const int yy = (tid >> 3) & 7, xx = tid & 7, zq = tid >> 6;
    const float* __restrict__ vol = hg.vol_r + sp.sample * hg.vol_bstride;
    float* img = hg.img_w + sp.sample * hg.img_bstride;
    const int oy = sp.ty * kT + yy;
    // (the tile loop's stores to the voxels redone here must have left this wave before they are overwritten)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef EDHIP_K1_STATS
    if (missed && (tid & 63) == 0)
        atomicAdd(&g_k1_stats[0], 1ull);
#endif
    for (int ti = 0; ti < sp.ntile; ++ti) {
        const int flags = uni(rec[ti].flags);
        const bool whole = (flags & kTUnfit) != 0;
        if (!whole && !missed)
            continue;        // (every tile the loops took: also a general tile without a box -- all samples constant)
        const int ox = (sp.tx0 + ti) * kT + xx;
        XEnt xe;
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            const int zi = zq + 4 * i;
            const int oz = sp.tz * kT + zi;
            if (oz >= hg.out_len[0] || oy >= hg.out_len[1] || ox >= hg.out_len[2])
                continue;
            k1_xent(smem, ti * kT + xx, (zi * kT + yy) * 32, xe);
            double d[3];
            k1_disp<0>(smem, xe, d);
            const int b[3] = {oz + hg.off[0], oy + hg.off[1], ox + hg.off[2]};
            double P[3] = {0.0, 0.0, 0.0};
            if (AFFINE) {
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    P[h] = fma(hp->affine[h * 4 + 2], (double)ox,
                               fma(hp->affine[h * 4 + 0], (double)oz,
                                   fma(hp->affine[h * 4 + 1], (double)oy, hp->affine[h * 4 + 3] + hp->offd[h])));
            }
            int st[3], raw[3];
            float fr[3];
            const bool cst = k1_coords<ORDER, AFFINE>(hg, hp, d, b, P, st, fr, raw);
            if (!whole) {
                // A fast tile's voxels worked with the RAW window start (no range test), a general tile's with the
                // mapped one; the loop served the voxel iff that window lay inside the box (a constant voxel of a
                // general tile needs no window).  The box of a fast tile holds only windows of coordinates inside
                // the array (see the kernel), so raw == mapped for every voxel of a fast tile skipped here.
                const bool gen = (flags & kTGen) != 0;
                if (gen && cst)
                    continue;
                const int rz = (gen ? st[0] : raw[0]) - rec[ti].b0[0], ry = (gen ? st[1] : raw[1]) - rec[ti].b0[1],
                          rx = (gen ? st[2] : raw[2]) - rec[ti].b0[2];
                const bool inside = rz >= 0 && rz + NT <= rec[ti].ext[0] && ry >= 0 && ry + NT <= rec[ti].ext[1] &&
                                    rx >= 0 && rx + NT + kPadX <= rec[ti].ext[2];
                if (inside)
                    continue;
            }
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
#ifdef EDHIP_K1_STATS
            atomicAdd(&g_k1_stats[whole ? 2 : 1], 1ull);
#endif
            for (long long ss = 0; ss < hg.nsteps; ++ss) {
                long long vol_off = 0, img_off = 0;
                if (hg.nstep)
                    k1_step_offsets(hp, ss, vol_off, img_off);
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
                store_out<OUT16>(img, img_off + obase, val, io16);
            }
        }
    }
}

// ================================================================================================
// the kernel
// ================================================================================================
// per-voxel state handed from the coordinate pass of tile t + 1 to its gather, one tile later
struct VoxState {
    int addr[2];          // LDS byte address of tap (0, 0, 0) in the copy that matches the window's parity
    float frac[2][3];
    int flg;              // general tiles: bit i = voxel i is gathered, bit 2 + i = voxel i is stored
};

template <int ORDER, bool AFFINE, bool OUT16, int SPLIT>
__global__ __launch_bounds__(kBlock, 4) void k1_fwd_kernel(const HotGeom hg)
{
    constexpr int NT = ORDER + 1;
    constexpr int kPadX = NT & 1;          // even orders read one zero-weight padding tap
    constexpr int NTX = NT + kPadX;
    constexpr int H = ORDER / 2;
    constexpr int ROW1 = 4 * kT * 32;      // Q row of the lane's second voxel (z + 4): an immediate
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int io16 = OUT16 ? hg.io16 : 0;
    K1Strip sp;
    if (!k1_strip(hg, sp, blockIdx.x))
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // profiling build, EDHIP_DEBUG_PTR: per-wave cycle sums of the loop's intervals (tools/k1_phases.py)
#ifdef EDHIP_EXPERIMENTS
    long long tacc[7] = {0, 0, 0, 0, 0, 0, 0};
    long long tmark = hg.dbgbuf ? (long long)__builtin_readcyclecounter() : 0;
#define ED_TICK(K) do { if (hg.dbgbuf) { const long long now_ = __builtin_readcyclecounter(); tacc[K] += now_ - tmark; tmark = now_; } } while (0)
#else
#define ED_TICK(K) do { } while (0)
#endif
    k1_prologue(hg, sp, smem, tid);

    const HotParams* hp = reinterpret_cast<const HotParams*>(smem + kK1Hot);
    TileRec* rec = reinterpret_cast<TileRec*>(smem + kK1Rec);
    const int boxbase = hg.off_box;                        // LDS byte offset of the first copy
    const int odd_shift = (hg.box_cap - 1) * 4;            // first copy -> second copy, one element back
    float* box0 = reinterpret_cast<float*>(smem + hg.off_box);
    float* box1 = box0 + hg.box_cap;       // cap = 56 (mod 64): the two copies sit on disjoint banks
    const int ntile = uni(sp.ntile);

    // per-lane values that stay fixed along the strip
    const int yy = lane >> 3, xx = lane & 7;
    const int oy = sp.ty * kT + yy;
    const int oz0 = sp.tz * kT + wave;                     // second voxel: + 4
    const int lrow = (wave * kT + yy) * 32;                // the lane's Q row inside a control column (second: + ROW1)
    const int obase = oz0 * hg.img_sz + oy * hg.img_sy + sp.tx0 * kT + xx;      // second: + 4 * img_sz
    const float* __restrict__ vol = hg.vol_r + sp.sample * hg.vol_bstride;
    float* img = hg.img_w + sp.sample * hg.img_bstride;
    double Pzy[3][2];      // affine: A[h][0] oz + A[h][1] oy + A[h][3] + off_h
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            Pzy[h][i] = AFFINE ? fma(hp->affine[h * 4 + 0], (double)(oz0 + 4 * i),
                                     fma(hp->affine[h * 4 + 1], (double)oy, hp->affine[h * 4 + 3] + hp->offd[h]))
                               : 0.0;

    // ---- the boxes of the strip's tiles -----------------------------------------------------------------------
    // wave t samples tile t: the coordinate at 4 x 4 x 4 of its voxels, widened by the margin
    for (int t = wave; t < ntile; t += 4) {
        const int nz = min(kT, hg.out_len[0] - sp.tz * kT), ny = min(kT, hg.out_len[1] - sp.ty * kT),
                  nx = min(kT, hg.out_len[2] - (sp.tx0 + t) * kT);
        const int pz = ((lane >> 4) * (nz - 1)) / 3, py = (((lane >> 2) & 3) * (ny - 1)) / 3, px = ((lane & 3) * (nx - 1)) / 3;
        XEnt xe;
        k1_xent(smem, t * kT + px, (pz * kT + py) * 32, xe);
        double d[3];
        k1_disp<0>(smem, xe, d);
        const int o[3] = {sp.tz * kT + pz, sp.ty * kT + py, (sp.tx0 + t) * kT + px};
        double c[3];
        int lo[3], hi[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            if (AFFINE)
                c[h] = fma(hp->affine[h * 4 + 2], (double)o[2],
                           fma(hp->affine[h * 4 + 0], (double)o[0],
                               fma(hp->affine[h * 4 + 1], (double)o[1], hp->affine[h * 4 + 3] + hp->offd[h]))) + d[h];
            else
                c[h] = (double)(o[h] + hg.off[h]) + d[h];
            const double cr = (ORDER & 1) ? c[h] : c[h] + 0.5;
            lo[h] = (int)floor(cr - hp->slack[h]);
            hi[h] = (int)floor(cr + hp->slack[h]);
        }
        wave_box63(lo, hi);
        // fast: a full tile whose every coordinate, margin included, is one coord_axis_fast accepts
        int rlo[3], rhi[3];
        bool fast = nz == kT && ny == kT && nx == kT && !ED_DBG(hg.dbg, 1 << 16);
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            rlo[h] = __builtin_amdgcn_readlane(lo[h], 63);
            rhi[h] = __builtin_amdgcn_readlane(hi[h], 63);
            fast = fast && rlo[h] >= ((ORDER & 1) ? 0 : 1) && rhi[h] <= hg.in_len[h] - 2;
        }
        if (!fast) {
            // general tile: the range of the MAPPED coordinate over the samples (a sample that maps to the constant
            // has no window); where the raw range straddles an end of the array, the end itself is included --
            // the fold of 'mirror' / 'reflect', the clamp of 'nearest'.  ('wrap' jumps to the far end: such a
            // box does not fit, and k1_fix or the spill levels take the tile.)
            bool cst = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                double m = c[h];
                if (!(m >= 0.0 && m <= hp->last[h]))
                    m = map_coordinate_fast(m, hg.in_len[h], hg.mode, hp->period[h], hp->inv_period[h]);
                cst = cst || !(m > -1.0);
                const double mr = (ORDER & 1) ? m : m + 0.5;
                lo[h] = (int)floor(mr - hp->slack[h]);
                hi[h] = (int)floor(mr + hp->slack[h]);
            }
            if (cst) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    lo[h] = 0x7fffffff;
                    hi[h] = (int)0x80000000;
                }
            }
            wave_box63(lo, hi);
        }
        if (lane == 63) {
            int blo[3], bhi[3];
            bool any = true;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                if (!fast) {
                    any = any && hi[h] >= lo[h];
                    // the lowest / highest floor a coordinate that stays in (or is folded / clamped back into) the
                    // array can have, wherever the raw range reaches beyond what coord_axis_fast accepts
                    const int lowfold = hg.mode == EDHIP_MODE_REFLECT ? -1 : 0;
                    if (rlo[h] < ((ORDER & 1) ? 0 : 1) && rhi[h] >= lowfold)
                        lo[h] = min(lo[h], lowfold);
                    if (rhi[h] > hg.in_len[h] - 2 && rlo[h] <= hg.in_len[h] - 1)
                        hi[h] = max(hi[h], hg.in_len[h] - 1);
                }
                blo[h] = lo[h] - H;
                bhi[h] = hi[h] - H + ORDER + (h == 2 ? kPadX : 0);
            }
            if (!any) {          // every sample maps to the constant
                blo[0] = blo[1] = blo[2] = 0;
                bhi[0] = bhi[1] = bhi[2] = -1;
            }
            k1_derive(hg, sp, rec, t, blo, bhi, kPadX, fast);
        }
    }
    lds_barrier();
    ED_TICK(0);

    // ---- staging: the source box of tile ti into LDS, two copies, the second shifted by one element -----------
    auto stage = [&](int ti, const float* src) {
        const int4 r0 = *reinterpret_cast<const int4*>(rec + ti);
        const int4 r1 = *(reinterpret_cast<const int4*>(rec + ti) + 1);
        const int4 r2 = *(reinterpret_cast<const int4*>(rec + ti) + 2);
        const int flags = uni(r0.w);
        if (!(flags & kTStaged) || ED_DBG(hg.dbg, 1 << 18))       // (ablation 1 << 18: no staging)
            return;
        const int by = uni(r1.y), pitch = uni(r1.w);
        if (flags & kTDma) {
            // LDS-DMA, plane by plane (planes dealt to the four waves): one wave-instruction fills 1 KiB = RW
            // consecutive rows of a plane (16 rows of 64 bytes, or 5 rows of 192 bytes with lanes 60-63 idle);
            // lane -> (row, 16-byte chunk).  The plane's address is scalar arithmetic; a lane adds its own row /
            // chunk offset, which only depends on the pitch.  (Rows in one flat sequence over the planes fill
            // every lane, but cost a division, two multiplies and 64-bit adds per lane and instruction.)
            const int ez = uni(r1.x);
            const bool p16 = pitch == 16;
            const int RW = p16 ? 16 : 5;
            const int lr = p16 ? lane >> 2 : (lane * 21846) >> 18;      // lane / 12
            const int q = p16 ? lane & 3 : lane - lr * 12;
            if (flags & kTZYin) {
                const float* g0 = src + uni(r2.z);
                const long long rowoff = (long long)lr * hg.vol_sy + 4 * q;
                for (int zr = wave; zr < ez; zr += 4) {
                    const float* gp = g0 + (long long)zr * hg.vol_sz;
                    const int lrow0 = zr * by;
                    for (int y0 = 0; y0 < by; y0 += RW) {
                        if (lr < RW && y0 + lr < by) {
                            const float* g = gp + ((long long)y0 * hg.vol_sy + rowoff);
                            glds16(g, box0 + (lrow0 + y0) * pitch);
                            glds16(g + 1, box1 + (lrow0 + y0) * pitch);
                        }
                    }
                }
            } else {
                // planes / rows beyond the array's z / y ends: the mirror map of the reference's taps
                // (deform.c:791-813), applied to the plane / row index
                const int b0z = uni(r0.x), b0y = uni(r0.y);
                const float* g0 = src + (uni(r0.z) + 4 * q);
                for (int zr = wave; zr < ez; zr += 4) {
                    const float* gp = g0 + (long long)mirror_i32(b0z + zr, hg.in_len[0]) * hg.vol_sz;
                    const int lrow0 = zr * by;
                    for (int y0 = 0; y0 < by; y0 += RW) {
                        if (lr < RW && y0 + lr < by) {
                            const float* g = gp + (long long)mirror_i32(b0y + y0 + lr, hg.in_len[1]) * hg.vol_sy;
                            glds16(g, box0 + (lrow0 + y0) * pitch);
                            glds16(g + 1, box1 + (lrow0 + y0) * pitch);
                        }
                    }
                }
            }
        } else {
            // the box sticks out along x: every box index goes through the mirror map, as the reference does with
            // the taps of a window that sticks out (deform.c:791-813)
            const int b0z = uni(r0.x), b0y = uni(r0.y), b0x = uni(r0.z), ex = uni(r1.z), nrows = uni(r2.x);
            const float inv_by = __frcp_rn((float)by);
            const bool xin = (flags & kTXin) != 0;
            const int sub = tid & 7;
            for (int r = tid >> 3; r < nrows; r += kBlock / 8) {
                const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                const int zs = mirror_i32(b0z + zr, hg.in_len[0]);
                const int ys = mirror_i32(b0y + yr, hg.in_len[1]);
                const float* rowp = src + (zs * hg.vol_sz + ys * hg.vol_sy);
                float* d0 = box0 + r * pitch;
                float* d1 = box1 + r * pitch;
                for (int xi = sub; xi < ex; xi += 8) {
                    const int xs = xin ? b0x + xi : mirror_i32(b0x + xi, hg.in_len[2]);
                    const float val = rowp[xs];
                    d0[xi] = val;
                    if (xi > 0)
                        d1[xi - 1] = val;
                }
            }
        }
    };
    // one voxel from the staged box
    auto gather = [&](int addr, const float (&frac)[3], int pitch, int plane) -> float {
        // (the fractions go through an empty asm statement: left alone, the compiler computes the weights of both
        // voxels in front of the first gather and holds 24 registers across it)
        float f0 = frac[0], f1 = frac[1], f2 = frac[2];
        asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2));
        float w0[NT], w1[NT], w2[NTX];
        weights_from_frac<float, ORDER>(f0, w0);
        weights_from_frac<float, ORDER>(f1, w1);
        weights_from_frac<float, ORDER>(f2, w2);
        if (kPadX)
            w2[NT] = 0.f;
        const float* bp = reinterpret_cast<const float*>(smem + addr);
        return pitch == 16 ? k1_gather<ORDER, 16, SPLIT>(bp, plane, w0, w1, w2) : k1_gather<ORDER, 48, SPLIT>(bp, plane, w0, w1, w2);
    };

    // ---- coordinates of the lane's two voxels of tile ti, as LDS addresses relative to the tile's box -------------
    // GEN false (fast tiles): no range test, no boundary map, every voxel gathered and stored.  GEN true: general
    // coordinates (deform.c:771-824), constant and valid flags.  Either way a window that is not inside the box
    // raises the lane's flag.
    int bad = 0;
    auto tile_coords = [&](auto gen_tag, int ti, VoxState& vs) {
        constexpr bool GEN = decltype(gen_tag)::value;
        const int4 r0 = *reinterpret_cast<const int4*>(rec + ti);
        const int4 r1 = *(reinterpret_cast<const int4*>(rec + ti) + 1);
        const int4 r2 = *(reinterpret_cast<const int4*>(rec + ti) + 2);
        const int b0z = uni(r0.x), b0y = uni(r0.y), b0x = uni(r0.z);
        const int pitch = uni(r1.w), plane = uni(r2.y);
        const int ez = uni(r1.x) - NT, ey = uni(r1.y) - NT, ex = uni(r1.z) - NTX;
        const int ox = (sp.tx0 + ti) * kT + xx;
        XEnt xe;
        k1_xent(smem, ti * kT + xx, lrow, xe);
        vs.flg = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double d[3];
            if (i == 0)
                k1_disp<0>(smem, xe, d);
            else
                k1_disp<ROW1>(smem, xe, d);
            if (ED_DBG(hg.dbg, 1 << 19)) {     // (sensitivity: the displacement -- table reads + fp64 sums -- a second time)
                XEnt x2 = xe;
                asm volatile("" : "+v"(x2.qa[0]), "+v"(x2.qa[1]), "+v"(x2.qa[2]), "+v"(x2.qa[3]));
                double d2[3];
                if (i == 0)
                    k1_disp<0>(smem, x2, d2);
                else
                    k1_disp<ROW1>(smem, x2, d2);
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    d[h] = (d[h] + d2[h]) * 0.5;
            }
            ED_NO_DS_MERGE();          // (the backend pairs the two voxels' reads into ds_read2st64_b64: 16 LDS cycles each)
            int rz, ry, rx;
            if constexpr (GEN) {
                const int b[3] = {oz0 + 4 * i + hg.off[0], oy + hg.off[1], ox + hg.off[2]};
                double P[3] = {0.0, 0.0, 0.0};
                if (AFFINE) {
#pragma unroll
                    for (int h = 0; h < 3; ++h)
                        P[h] = fma(hp->affine[h * 4 + 2], (double)ox, Pzy[h][i]);
                }
                int start[3];
                const bool cst = k1_coords<ORDER, AFFINE>(hg, hp, d, b, P, start, vs.frac[i]);
                const bool valid = oz0 + 4 * i < hg.out_len[0] && oy < hg.out_len[1] && ox < hg.out_len[2];
                rz = start[0] - b0z;
                ry = start[1] - b0y;
                rx = start[2] - b0x;
                if (valid && !cst)
                    bad |= (rz | ry | rx) | ((ez - rz) | (ey - ry) | (ex - rx));
                vs.flg |= (valid && !cst ? 1 << i : 0) | (valid ? 4 << i : 0);
            } else {
                // window start relative to the box = floor part of the coordinate + k (tile- and lane-constant)
                const int kz = AFFINE ? -H - b0z : oz0 + 4 * i + hg.off[0] - H - b0z;
                const int ky = AFFINE ? -H - b0y : oy + hg.off[1] - H - b0y;
                const int kx = AFFINE ? -H - b0x : ox + hg.off[2] - H - b0x;
                int ci[3];
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    // (coord_axis_fast without its range test: the same floor, the same fraction)
                    const double c = AFFINE ? fma(hp->affine[h * 4 + 2], (double)ox, Pzy[h][i]) + d[h] : d[h];
                    const double fl = floor((ORDER & 1) ? c : c + 0.5);
                    ci[h] = (int)fl;
                    vs.frac[i][h] = (float)(c - fl);
                }
                rz = ci[0] + kz;
                ry = ci[1] + ky;
                rx = ci[2] + kx;
                bad |= (rz | ry | rx) | ((ez - rz) | (ey - ry) | (ex - rx));
            }
            // (24-bit multiplies: a window inside the box has small non-negative offsets; one outside is flagged)
            const int off = __mul24(rz, plane) + (__mul24(ry, pitch) + rx);
            // aligned pairs from the copy whose shift matches the parity of rx (pitch and plane are even)
            vs.addr[i] = __mul24(off & 1, odd_shift) + (off * 4 + boxbase);
        }
    };

    // ---- tile loop, software-pipelined: the coordinates of the next tile of the class under the copies of this one ---
    auto run_tiles = [&](auto gen_tag) {
        constexpr bool GEN = decltype(gen_tag)::value;
        auto next_tile = [&](int ti) {          // the next tile of the class after ti (ntile: none)
            int t = ti + 1;
            for (; t < ntile; ++t) {
                const int f = uni(rec[t].flags);
                if (GEN ? ((f & kTGen) && !(f & kTUnfit)) : (f & kTFast) != 0)
                    break;
            }
            return t;
        };
        VoxState cur, nxt;
        int ti = next_tile(-1);
        if (ti < ntile)
            tile_coords(gen_tag, ti, cur);
        ED_TICK(GEN ? 6 : 1);
        while (ti < ntile) {
            long long vol_off = 0, img_off = 0;
            if (hg.nstep)
                k1_step_offsets(hp, 0, vol_off, img_off);
            stage(ti, vol + vol_off);
            ED_TICK(GEN ? 6 : 2);
            const int tn = next_tile(ti);
            if (tn < ntile)
                tile_coords(gen_tag, tn, nxt);
            ED_TICK(GEN ? 6 : 3);
            const bool staged = !GEN || (uni(rec[ti].flags) & kTStaged);
            const int pitch = uni(rec[ti].pitch), plane = uni(rec[ti].plane);
            for (long long ss = 0; ss < hg.nsteps; ++ss) {
                if (ss > 0) {
                    k1_step_offsets(hp, ss, vol_off, img_off);
                    stage(ti, vol + vol_off);
                }
                if (staged)
                    dma_barrier();       // B2: retires this wave's copies (vmcnt) and everyone's
                ED_TICK(GEN ? 6 : 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (GEN && !(cur.flg & (4 << i)))
                        continue;
                    float val = hg.cval;
                    if (ED_DBG(hg.dbg, 1 << 17))           // (ablation: no gather)
                        val = cur.frac[i][0] + cur.frac[i][1] + cur.frac[i][2] + __int_as_float(cur.addr[i]);
                    else if (!GEN || (cur.flg & (1 << i))) {
                        val = gather(cur.addr[i], cur.frac[i], pitch, plane);
                        if (ED_DBG(hg.dbg, 1 << 23)) {     // (sensitivity: weights + gather a second time)
                            int a2 = cur.addr[i];
                            asm volatile("" : "+v"(a2));
                            val = (val + gather(a2, cur.frac[i], pitch, plane)) * 0.5f;
                        }
                    }
                    if (ED_DBG(hg.dbg, 1 << 20) && val != -12345.678f)      // (ablation: no stores)
                        continue;
                    // streaming store (a tile writes 32-byte row segments)
                    store_out<OUT16>(img, img_off + obase + i * 4 * hg.img_sz + ti * kT, val, io16);
                }
                if (staged)
                    lds_barrier();       // B1: every gather of this tile is done with the box
            }
            ED_TICK(GEN ? 6 : 5);
            cur = nxt;
            ti = tn;
        }
    };
    run_tiles(std::false_type{});
    bool anygen = false, unfit = false;
    for (int t = 0; t < ntile; ++t) {
        const int f = uni(rec[t].flags);
        anygen = anygen || (f & kTGen);
        unfit = unfit || (hg.self_serve && (f & kTUnfit));
    }
    if (anygen)
        run_tiles(std::true_type{});
#ifdef EDHIP_EXPERIMENTS
    if (hg.dbgbuf && lane == 0) {
        unsigned long long* d = hg.dbgbuf + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int q = 0; q < 7; ++q)
            d[q] = (unsigned long long)tacc[q];
        d[7] = (unsigned long long)ntile;
    }
#endif
#undef ED_TICK
    // ---- what the loops could not serve: unfit tiles (self-serve), windows outside a sampled box ------------------
    const bool missed = __any(bad < 0);
    if (unfit || missed)
        k1_fix<ORDER, AFFINE, OUT16>(hg, sp, smem, missed, io16);
}

template <int ORDER>
hipError_t launch_k1_order(const HotGeom& hg, unsigned nblk, size_t lds, hipStream_t stream)
{
    // reads of the gather: 1 = in order of use, kept apart (shipped); the profiling build also has 0 = as the backend
    // fuses them (ds_read2_b64: +25 %).  (Two rows / a plane at a time in reverse order of use -- one s_waitcnt per
    // group instead of one per read, but a group's first multiply-add waits for its last read -- measured +16 %,
    // profiles/r05_ablate_k1.txt; that variant lived until commit 7c13be2.)
    [[maybe_unused]] const int split = ed_env("EDHIP_K1_SPLIT") ? atoi(ed_env("EDHIP_K1_SPLIT")) : 1;
#ifdef EDHIP_EXPERIMENTS
#define ED_K1_GO(A, O16)                                                                                              \
    do {                                                                                                              \
        if (split != 0)                                                                                               \
            hipLaunchKernelGGL((k1_fwd_kernel<ORDER, A, O16, 1>), dim3(nblk), dim3(kBlock), lds, stream, hg);         \
        else                                                                                                          \
            hipLaunchKernelGGL((k1_fwd_kernel<ORDER, A, O16, 0>), dim3(nblk), dim3(kBlock), lds, stream, hg);         \
    } while (0)
#else
#define ED_K1_GO(A, O16) hipLaunchKernelGGL((k1_fwd_kernel<ORDER, A, O16, 1>), dim3(nblk), dim3(kBlock), lds, stream, hg)
#endif
    if (hg.io16) {
        if (hg.has_affine)
            ED_K1_GO(true, true);
        else
            ED_K1_GO(false, true);
    } else if (hg.has_affine) {
        ED_K1_GO(true, false);
    } else {
        ED_K1_GO(false, false);
    }
#undef ED_K1_GO
    return hipGetLastError();
}

}  // namespace

// LDS: x table | tile records | reduction slots | parameters | 64 Q rows | box pair.  Returns 0 when the
// control grid is too wide for a useful box.
size_t k1_lds_bytes(int ncpx, int* box_cap, int* off_box, bool large)
{
    const size_t q = (size_t)kQCol * (size_t)ncpx;
    const size_t off = (kK1Q + q + 15) & ~(size_t)15;
    *off_box = (int)off;
    // four workgroups per CU -> 40960 bytes each; large boxes (three per CU) for strongly deformed volumes
    size_t budget = large ? 52 * 1024 : 40 * 1024;
    if (const char* kb = ed_env("EDHIP_HOT_FWD_KB"))      // experiment: fewer workgroups per CU
        budget = (size_t)atoi(kb) * 1024;
    if (off + 2 * 4 * 2488 > budget)
        budget = 64 * 1024;
    if (off + 2 * 4 * 2488 > budget)
        return 0;
    size_t cap = (budget - off) / 8;
    cap = ((cap - 56) / 64) * 64 + 56;        // cap = 56 (mod 64): the copies sit on disjoint banks
    *box_cap = (int)cap;
    return off + 2 * 4 * cap;
}

#ifdef EDHIP_K1_STATS
// read (and clear) the K1 counters
extern "C" int edhip_debug_k1_stats(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_k1_stats), sizeof(z)) != hipSuccess)
        return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k1_stats), z, sizeof(z)) != hipSuccess;
}
#endif

hipError_t launch_k1_level1(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream)
{
    if (hg.strip_tiles > kK1Strip)
        return hipErrorNotSupported;
    switch (order) {
    case 1: return launch_k1_order<1>(hg, nblk, lds, stream);
    case 2: return launch_k1_order<2>(hg, nblk, lds, stream);
    case 3: return launch_k1_order<3>(hg, nblk, lds, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace tile
}  // namespace ed
