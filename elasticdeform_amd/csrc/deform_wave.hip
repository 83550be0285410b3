// deform_wave.hip -- K1 (forward gather) and K2 (gradient scatter-add) for float32 volumes with 3
// deformed axes and unit x-stride, spline orders 1-5: ONE WAVEFRONT PER OUTPUT TILE, no workgroup
// barriers.  A workgroup is a single wave; it owns its 8 x 8 x 8 output tile, the tile's source box
// in LDS and nothing else, so the only synchronisation left is s_waitcnt inside the wave, and the
// 12 waves of a CU drift apart: while one gathers (LDS), others compute coordinates (fp64 VALU) or
// wait for their staging copies.
//
// Per-voxel pipeline of DeformGrid's hot loop (deform.c:649-924 forward, :926-997 gradient):
//
//   lane map   lane = 8 y + z owns one (z, y) row of the tile and walks its 8 voxels along x.
//              * the displacement spline's row of control columns Q[oz][oy][.] (tile_tables_kernel)
//                is lane-constant: the 4 columns x 3 components a voxel needs stay in 24 VGPRs for
//                the whole strip and are reloaded only when the walk crosses a control interval;
//              * the x table entry (cubic weights, control indices: the reference's dsplvals,
//                deform.c:639-647) is wave-uniform: scalar loads, SGPR operands of the 12 fp64 FMAs.
//              Coordinates therefore cost no LDS access at all (the 4-wave kernels read 12 table
//              taps per voxel from LDS).
//   box        pass 1 walks the row once for the tile's bounding box of tap windows (DPP reduction,
//              v_readlane -- no LDS atomics, no slots), pass 2 recomputes the coordinates (same
//              instructions, same inputs: bit-identical) right before each voxel's taps: nothing is
//              kept per voxel, so both passes are short rolled loops.
//   K1 staging one copy of the box, row pitch 16 floats, plane stride 16 rows + 2 floats: with lanes on
//              64 different (z, y) rows of the box the aligned 8-byte reads of a 32-lane group fall on
//              32 different bank pairs when the deformation is rigid (tools/sim/conflicts_walk_x.py:
//              176 LDS cycles per 64 voxels on the cfg2 field against 124 for two shifted copies,
//              which need twice the LDS and twice the staging traffic).  A window of 4 taps at either
//              parity is covered by three aligned ds_read_b64; the tap outside the window gets weight
//              zero AND is replaced by zero (so an Inf / NaN next to the window cannot leak in).
//              Interior boxes are staged with LDS-DMA (global_load_lds_dwordx4, one instruction per
//              box plane); boxes that cross the x ends of the volume are mirror-mapped per element,
//              as the reference does with the taps of a window that sticks out (deform.c:791-813).
//   K1 gather  contracted over (z, y) first per box column, then over x: (NT + 1) NT (NT + 1) + NT + 1
//              FMAs per voxel (105 for order 3).
//   adaptive   a tile whose box exceeds the wave's LDS is processed as two 8 x 8 x 4 halves along x
//              (same lanes, half the walk); what still does not fit goes to the spill list and the
//              general kernels (deform_tile.hip).
//
// The sliding-register-window variant (keep a lane's 4 x 4 x 4 window in registers and fetch only
// the column that enters when start_x advances) was evaluated on the cfg2 field before writing this
// kernel (tools/sim/sliding_window_stats.py): 78 % of a lane's steps keep (start_z, start_y) and
// advance start_x by one, but only 2 % of a WAVE's steps do so in all 64 lanes (0.3 % at sigma 10),
// so a wave would execute the incremental path and the full reload on almost every step.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ed_device.h"
#include "ed_params.h"
#include "ed_tile.h"

namespace ed {
namespace tile {

namespace {

typedef int int8v __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// min of lo[3] / max of hi[3] over the wave, result wave-uniform (six interleaved DPP chains ending
// in lane 63, see deform_hot.hip's box_reduce_to_lds for why this is hand-written)
#define ED_RED6(CTRL)                                      \
    "v_min_i32_dpp %0, %0, %0 " CTRL "\n\t"                \
    "v_min_i32_dpp %1, %1, %1 " CTRL "\n\t"                \
    "v_min_i32_dpp %2, %2, %2 " CTRL "\n\t"                \
    "v_max_i32_dpp %3, %3, %3 " CTRL "\n\t"                \
    "v_max_i32_dpp %4, %4, %4 " CTRL "\n\t"                \
    "v_max_i32_dpp %5, %5, %5 " CTRL "\n\t"
__device__ __forceinline__ void wave_box(int (&lo)[3], int (&hi)[3])
{
    asm volatile("s_nop 1\n\t"
                 ED_RED6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 ED_RED6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_half_mirror row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_mirror row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 ED_RED6("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        lo[h] = __builtin_amdgcn_readlane(lo[h], 63);
        hi[h] = __builtin_amdgcn_readlane(hi[h], 63);
    }
}
#undef ED_RED6

__device__ __forceinline__ void glds16(const float* g, float* lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

struct WaveStrip {
    int tz, ty, tx0, ntile, sample;
};

// strips are dealt to the 8 XCDs in contiguous chunks (block b runs on XCD b % 8): neighbouring
// strips, whose source boxes overlap, share an L2
__device__ __forceinline__ void strip_decode(const HotGeom& hg, WaveStrip& sp, int s)
{
    sp.sample = s / hg.nstrips;
    s -= sp.sample * hg.nstrips;
    const int sx = s % hg.strips_x;
    s /= hg.strips_x;
    sp.ty = s % hg.tiles[1];
    sp.tz = s / hg.tiles[1];
    sp.tx0 = sx * hg.strip_tiles;
    sp.ntile = min(hg.strip_tiles, hg.tiles[2] - sp.tx0);
}
__device__ __forceinline__ bool wave_strip(const HotGeom& hg, WaveStrip& sp, int b)
{
    const int per = (hg.total_strips + 7) >> 3;
    const int s = (b & 7) * per + (b >> 3);
    if ((b >> 3) >= per || s >= hg.total_strips)
        return false;
    strip_decode(hg, sp, s);
    return true;
}
// The lane's control columns of its Q row: 4 columns x 3 components (fp64), reloaded when the
// wave-uniform index tuple of the x table changes (once per control interval).
struct QCols {
    double v[4][3];
    int idx[4];          // wave-uniform: the element offsets these columns were loaded from
};

__device__ __forceinline__ void qcols_load(QCols& qc, const double* __restrict__ qrow, const int (&idx)[4])
{
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const double2 a = *reinterpret_cast<const double2*>(qrow + idx[l]);
        qc.v[l][0] = a.x;
        qc.v[l][1] = a.y;
        qc.v[l][2] = qrow[idx[l] + 2];
        qc.idx[l] = idx[l];
    }
    // Wait HERE, on the rare path.  Otherwise the compiler puts s_waitcnt vmcnt(0) at the join in front
    // of every voxel's coordinates, and there it also waits for the output stores / flush atomics this
    // wave has in flight (vmcnt counts those too): a memory round trip per voxel.
    __builtin_amdgcn_s_waitcnt(0x0F70);
}

// Phase A for one voxel (deform.c:649-824): displacement from the lane's control columns with the
// wave-uniform cubic weights `tw`, (affine), + offset, window start and fractional offsets.  Same
// operations in the same order as hot_coords (deform_hot.hip) and voxel_coords (deform_tile.hip): a
// voxel gets bit-identical (start, frac) whichever kernel serves its tile.  `b[h]` = output index +
// crop offset along axis h (no affine) or 0 (affine: the real base is in `P`).  Returns true when the
// voxel maps to the constant.
template <int ORDER, bool AFFINE>
__device__ __forceinline__ bool wave_coords(const HotGeom& hg, const HotParams* hp, const QCols& qc,
                                            const double (&tw)[4], const int (&b)[3], const double (&P)[3],
                                            int* start, float* frac)
{
    double d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        d[c] = tw[0] * qc.v[0][c];
#pragma unroll
    for (int l = 1; l < 4; ++l)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            d[c] = fma(tw[l], qc.v[l][c], d[c]);
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

__device__ __forceinline__ void wave_step_offsets(const HotParams* hp, long long ss, long long& vol_off,
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

// Rarely used wave-uniform values (boundary-map periods, the affine map, step-axis strides) are parked
// in the head of the wave's LDS: as kernel arguments they are hoisted into scalar registers for the
// whole kernel, and the spills (v_writelane / v_readlane) cost more than the occasional broadcast read.
constexpr int kWaveHead = 416;
static_assert(sizeof(HotParams) <= kWaveHead, "HotParams must fit the head of the wave's LDS");
__device__ __forceinline__ void wave_prologue(const HotGeom& hg, char* smem, int lane)
{
    HotParams* hp = reinterpret_cast<HotParams*>(smem);
    if (lane < 12) {
        hp->affine[lane] = hg.affine[lane];
        if (lane < 3) {
            hp->offd[lane] = (double)hg.off[lane];
            hp->last[lane] = (double)(hg.in_len[lane] - 1);
            hp->period[lane] = hg.period[lane];
            hp->inv_period[lane] = hg.inv_period[lane];
        }
        if (lane < 8) {
            hp->step_len[lane] = hg.step_len[lane];
            hp->in_step_stride[lane] = hg.vol_step[lane];
            hp->out_step_stride[lane] = hg.img_step[lane];
        }
        if (lane == 0)
            hp->nstep = hg.nstep;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): one wave, DS operations are ordered
    asm volatile("" ::: "memory");
}

// Per-lane, per-strip constants of the row walk
struct RowWalk {
    int oz, oy;
    bool vzy;                 // the lane's (z, y) row exists in the output
    const double* qrow;       // the lane's row of control columns (clamped to the last row)
    double Pzy[3];            // affine: A[h][0] oz + A[h][1] oy + A[h][3] + off_h
};

// x table entry of output column ox (wave-uniform): weights and control column offsets
struct XEntry {
    double w[4];
    int idx[4];
};
__device__ __forceinline__ void xentry_load(const AxTab* __restrict__ xt, int ox, XEntry& e)
{
    const AxTab* t = xt + ox;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        e.w[l] = t->w[l];
        e.idx[l] = t->idx[l];
    }
}

// coordinates of the voxel at output column `ox` of the lane's row, given its x table entry
template <int ORDER, bool AFFINE>
__device__ __forceinline__ bool entry_voxel(const HotGeom& hg, const HotParams* hp, const RowWalk& rw, QCols& qc,
                                            const XEntry& xe, int ox, int* start, float* frac)
{
    if (xe.idx[0] != qc.idx[0] || xe.idx[1] != qc.idx[1] || xe.idx[2] != qc.idx[2] || xe.idx[3] != qc.idx[3])
        qcols_load(qc, rw.qrow, xe.idx);          // wave-uniform branch
    const int b[3] = {rw.oz + hg.off[0], rw.oy + hg.off[1], ox + hg.off[2]};
    double P[3] = {0.0, 0.0, 0.0};
    if (AFFINE) {
#pragma unroll
        for (int h = 0; h < 3; ++h)
            P[h] = fma(hp->affine[h * 4 + 2], (double)ox, rw.Pzy[h]);
    }
    return wave_coords<ORDER, AFFINE>(hg, hp, qc, xe.w, b, P, start, frac);
}
template <int ORDER, bool AFFINE>
__device__ __forceinline__ bool row_voxel(const HotGeom& hg, const HotParams* hp, const RowWalk& rw, QCols& qc,
                                          const AxTab* __restrict__ xt, int ox, int* start, float* frac)
{
    XEntry xe;
    xentry_load(xt, min(ox, hg.out_len[2] - 1), xe);
    return entry_voxel<ORDER, AFFINE>(hg, hp, rw, qc, xe, ox, start, frac);
}

// pass 1: bounding box of the tap windows of the lane's voxels [ox0, ox0 + nx), reduced over the wave.
// Two voxels per iteration (independent fp64 chains); the table entries of the next pair are requested
// (scalar loads) as soon as this pair's coordinates no longer need the registers.  nx is even.
template <int ORDER, bool AFFINE>
__device__ __forceinline__ void row_box(const HotGeom& hg, const HotParams* hp, const RowWalk& rw, QCols& qc,
                                        const AxTab* __restrict__ xt, int ox0, int nx, int (&lo)[3], int (&hi)[3])
{
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        lo[h] = 0x7fffffff;
        hi[h] = (int)0x80000000;
    }
    const int last = hg.out_len[2] - 1;
    XEntry ea, eb;
    xentry_load(xt, min(ox0, last), ea);
    xentry_load(xt, min(ox0 + 1, last), eb);
#pragma unroll 1
    for (int k = 0; k < nx; k += 2) {
        const int ox = ox0 + k;
        int sa[3], sb[3];
        float fa[3], fb[3];
        const bool ca = entry_voxel<ORDER, AFFINE>(hg, hp, rw, qc, ea, ox, sa, fa);
        const bool cb = entry_voxel<ORDER, AFFINE>(hg, hp, rw, qc, eb, ox + 1, sb, fb);
        xentry_load(xt, min(ox + 2, last), ea);
        xentry_load(xt, min(ox + 3, last), eb);
        if (rw.vzy && ox <= last && !ca) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                lo[h] = min(lo[h], sa[h]);
                hi[h] = max(hi[h], sa[h] + ORDER);
            }
        }
        if (rw.vzy && ox + 1 <= last && !cb) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                lo[h] = min(lo[h], sb[h]);
                hi[h] = max(hi[h], sb[h] + ORDER);
            }
        }
    }
    wave_box(lo, hi);
}

// ================================================================================================
// K1: forward
// ================================================================================================
// Box geometry of one (sub-)tile in LDS: element (rz, ry, rx) at rz * ps + ry * pitch + rx; the pitch
// is 16 floats for a whole tile and 12 where that is enough (the x-halves of a split tile): both are
// whole 16-byte chunks, which LDS-DMA needs, and put the (z, y) rows of a 32-lane group on different
// bank pairs together with the odd plane stride.
struct BoxLayout {
    int b0[3], ext[3];
    int pitch, ps;
    bool any, fits;
};
// z-planes of taps in flight per voxel (one plane is requested ahead where the registers allow it:
// three waves per SIMD, orders up to 3)
template <int ORDER, int OCC>
constexpr int gather_bufs() { return (ORDER <= 3 && OCC <= 3) ? 2 : 1; }

template <int ORDER>
__device__ __forceinline__ void fwd_layout(const int (&lo)[3], const int (&hi)[3], int box_cap, bool half,
                                           BoxLayout& bl)
{
    constexpr int NT = ORDER + 1;
    constexpr int NP = (NT + 2) / 2;          // aligned pairs that cover a window at either parity
    bl.any = hi[0] >= lo[0];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        bl.b0[h] = lo[h];
        bl.ext[h] = hi[h] - lo[h] + 1;
    }
    // columns the reads can touch: the last window starts at ext_x - NT, its first pair at that & ~1
    const int need = ((bl.ext[2] - NT) & ~1) + 2 * NP;
    bl.pitch = (half && need <= 12) ? 12 : 16;
    bl.ps = bl.ext[1] * bl.pitch + 2;         // planes one bank pair apart (see the header)
    bl.fits = bl.any && need <= 16 && (unsigned)bl.ext[0] <= 4096u && (unsigned)bl.ext[1] <= 4096u &&
              bl.ext[0] * bl.ps <= box_cap;
}

// stage the source box (one copy).  x-interior boxes: LDS-DMA, one instruction per 64 16-byte chunks
// of a plane, the (z, y) mirror map applied to the lane's row; otherwise element by element.
__device__ __forceinline__ void fwd_stage(const HotGeom& hg, const BoxLayout& bl, const float* __restrict__ src,
                                          float* box, int lane)
{
    const bool x_inside = bl.b0[2] >= 0 && bl.b0[2] + bl.pitch <= hg.in_len[2];
    const bool zy_inside = bl.b0[0] >= 0 && bl.b0[0] + bl.ext[0] <= hg.in_len[0] && bl.b0[1] >= 0 &&
                           bl.b0[1] + bl.ext[1] <= hg.in_len[1];
    if (x_inside) {
        const int cpr = bl.pitch >> 2;                 // 16-byte chunks per row: 4 or 3
        const int nchunk = bl.ext[1] * cpr;            // per plane
        for (int c0 = 0; c0 < nchunk; c0 += 64) {
            const int c = c0 + lane;
            const int row = cpr == 4 ? c >> 2 : (c * 21846) >> 16;      // c / 3 for c < 2^15
            const int ch = c - row * cpr;
            const bool live = c < nchunk;
            const int ys = zy_inside ? bl.b0[1] + row : mirror_i32(bl.b0[1] + min(row, bl.ext[1] - 1), hg.in_len[1]);
            const int rowoff = ys * hg.vol_sy + bl.b0[2] + 4 * ch;
            for (int rz = 0; rz < bl.ext[0]; ++rz) {
                const int zs = zy_inside ? bl.b0[0] + rz : mirror_i32(bl.b0[0] + rz, hg.in_len[0]);
                if (live)
                    glds16(src + (zs * hg.vol_sz + rowoff), box + rz * bl.ps + c0 * 4);
            }
        }
    } else {
        // every box index goes through the mirror map (deform.c:791-813); 16 lanes per row
        const int sub = lane & 15;
        const int nrows = bl.ext[0] * bl.ext[1];
        const float inv_by = 1.0f / (float)bl.ext[1];
        const int xs = mirror_i32(bl.b0[2] + sub, hg.in_len[2]);
        for (int r = lane >> 4; r < nrows; r += 4) {
            const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * bl.ext[1];
            const int zs = mirror_i32(bl.b0[0] + zr, hg.in_len[0]);
            const int ys = mirror_i32(bl.b0[1] + yr, hg.in_len[1]);
            if (sub < bl.ext[2])
                box[zr * bl.ps + yr * bl.pitch + sub] = src[zs * hg.vol_sz + ys * hg.vol_sy + xs];
        }
    }
}

// (ORDER + 1)^3 taps of one voxel from the staged box.  `bp` points at the aligned pair that holds
// tap (0, 0, 0), `par` is the parity of the window's x start.  A window of NT taps at either parity
// lies in NP aligned pairs; `wx` is the x weight vector shifted to the window's parity (zero outside)
// and the value outside the window is replaced by zero as well (an Inf / NaN next to the window must
// not leak in through 0 * Inf).  Accumulation order: x, then y, then z, each from zero -- the same
// sums, bit for bit, as the 4-wave kernels (hot_gather, deform_hot.hip), so a voxel gets the same value
// whichever level serves its tile (the crop identity full[crop] == cropped holds bit for bit).
// The reads of the next NBUF - 1 z-planes are in flight while a plane is accumulated; the order is
// pinned by hand (empty asm statements): left alone, the compiler sinks every FMA below the last read
// of the voxel and spills.
template <int ORDER, int NBUF, int PITCH>
__device__ __forceinline__ float wave_gather(const float* bp, int ps, bool par, const float (&w0)[ORDER + 1],
                                             const float (&w1)[ORDER + 1], const float (&w2)[ORDER + 1])
{
    constexpr int NT = ORDER + 1;
    constexpr int NP = (NT + 2) / 2;
    constexpr int NC = NT + 1;                 // box columns a window can touch (either parity)
    float wx[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const float wl = j > 0 ? w2[j - 1] : 0.f, wr = j < NT ? w2[j] : 0.f;
        wx[j] = par ? wl : wr;
    }
    float v[NBUF][NT][2 * NP];
    auto rd = [&](int l0) {
        const float* pp = bp + l0 * ps;
#pragma unroll
        for (int l1 = 0; l1 < NT; ++l1) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float2 pr = *reinterpret_cast<const float2*>(pp + l1 * PITCH + 2 * p);
                ED_NO_DS_MERGE();              // (ds_read2_b64 runs at half the rate of two ds_read_b64)
                v[l0 % NBUF][l1][2 * p] = pr.x;
                v[l0 % NBUF][l1][2 * p + 1] = pr.y;
            }
            if (2 * NP > NC)                   // keep the last pair an 8-byte read (64 banks; a 4-byte read has 32)
                asm volatile("" ::"v"(v[l0 % NBUF][l1][2 * NP - 1]));
        }
    };
    float a0 = 0.f;
    auto acc = [&](int l0) {
        float a1 = 0.f;
#pragma unroll
        for (int l1 = 0; l1 < NT; ++l1) {
            const float first = par ? 0.f : v[l0 % NBUF][l1][0];
            const float lastv = par ? v[l0 % NBUF][l1][NT] : 0.f;
            float a2 = 0.f;
#pragma unroll
            for (int j = 0; j < NC; ++j)
                a2 = fmaf(wx[j], j == 0 ? first : (j == NT ? lastv : v[l0 % NBUF][l1][j]), a2);
            a1 = fmaf(w1[l1], a2, a1);
        }
        a0 = fmaf(w0[l0], a1, a0);
        asm volatile("" : "+v"(a0));
    };
#pragma unroll
    for (int l0 = 0; l0 < NBUF - 1 && l0 < NT; ++l0)
        rd(l0);
#pragma unroll
    for (int l0 = 0; l0 < NT; ++l0) {
        if (l0 + NBUF - 1 < NT)
            rd(l0 + NBUF - 1);
        acc(l0);
    }
    return a0;
}

// OCC: waves per SIMD the kernel is compiled for (3: 168 VGPRs, 12.5 KiB of LDS per wave; 4: 128 VGPRs,
// 10 KiB)
template <int ORDER, bool AFFINE, int OCC>
__global__ __launch_bounds__(64, OCC) void wave_fwd_kernel(const HotGeom hg, const AxTab* __restrict__ xt,
                                                         const double* __restrict__ qtab,
                                                         const float* __restrict__ vol0, float* __restrict__ img0)
{
    constexpr int NT = ORDER + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem);
    float* box = reinterpret_cast<float*>(smem + kWaveHead);
    const int lane = threadIdx.x;
    WaveStrip sp;
#ifdef EDHIP_EXPERIMENTS
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
    if (!wave_strip(hg, sp, blockIdx.x))
        return;
    wave_prologue(hg, smem, lane);
    const int yy = lane >> 3, zz = lane & 7;
    {
    const float* __restrict__ vol = vol0 + sp.sample * hg.vol_bstride;
    float* __restrict__ img = img0 + sp.sample * hg.img_bstride;

    RowWalk rw;
    rw.oz = sp.tz * kT + zz;
    rw.oy = sp.ty * kT + yy;
    rw.vzy = rw.oz < hg.out_len[0] && rw.oy < hg.out_len[1];
    rw.qrow = qtab + sp.sample * hg.q_bstride +
              ((long long)min(rw.oz, hg.out_len[0] - 1) * hg.out_len[1] + min(rw.oy, hg.out_len[1] - 1)) * (4 * hg.ncpx);
#pragma unroll
    for (int h = 0; h < 3; ++h)
        rw.Pzy[h] = AFFINE ? fma(hp->affine[h * 4 + 0], (double)rw.oz,
                                 fma(hp->affine[h * 4 + 1], (double)rw.oy, hp->affine[h * 4 + 3] + hp->offd[h]))
                           : 0.0;
    const int obase = rw.oz * hg.img_sz + rw.oy * hg.img_sy;
    QCols qc;
    {
        XEntry xe;
        xentry_load(xt, min(sp.tx0 * kT, hg.out_len[2] - 1), xe);
        qcols_load(qc, rw.qrow, xe.idx);
    }

    for (int ti = 0; ti < sp.ntile; ++ti) {
        const int tile_id = sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 + ti;
        const int ox_tile = (sp.tx0 + ti) * kT;
        // whole tile first; if its box does not fit the wave's LDS, two halves along x
        int nsub = 1, nx = kT;
        bool beyond = false;       // spill feedback: the tile would not fit the standard boxes
        for (int sub = 0; sub < nsub; ++sub) {
            const int ox0 = ox_tile + sub * nx;
            int lo[3], hi[3];
            row_box<ORDER, AFFINE>(hg, hp, rw, qc, xt, ox0, nx, lo, hi);
            BoxLayout bl;
            fwd_layout<ORDER>(lo, hi, hg.box_cap, nsub == 2, bl);
            // (a whole tile that fits the large box only: its halves are taken as three quarters of it)
            if (bl.any)
                beyond = nsub == 1 ? (bl.fits && bl.ext[0] * bl.ps * 3 > hg.small_cap * 4)
                                   : (beyond || !bl.fits || bl.ext[0] * bl.ps > hg.small_cap);
            if (nsub == 1 && hg.boxes && lane < 6) {      // EDHIP_FLAG_KEEP_BOXES: the box goes to the gradient call too
                const int v = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lane == 2 ? lo[2] : lane == 3 ? hi[0]
                              : lane == 4 ? hi[1] : hi[2];
                hg.boxes[(size_t)tile_id * 8 + lane] = v;
            }
            if (bl.any && !bl.fits) {
                if (nsub == 1) {
                    nsub = 2;
                    nx = kT / 2;
                    sub = -1;           // start over with the halves
                    continue;
                }
                if (lane == 0) {        // hand the whole tile to the general kernels
                    const int slot = atomicAdd(&hg.spill[0], 1);
                    hg.spill[1 + slot] = tile_id;
                }
                break;
            }
            const int ps = bl.ps, pitch = bl.pitch;
            for (long long ss = 0; ss < hg.nsteps; ++ss) {
                long long vol_off = 0, img_off = 0;
                if (hg.nstep)
                    wave_step_offsets(hp, ss, vol_off, img_off);
#ifdef EDHIP_EXPERIMENTS
                if (!ED_DBG(hg.dbg, 2))             // ablation: no staging
#endif
                if (bl.any) {
                    // (the previous gather's reads have all returned: their values were stored)
                    fwd_stage(hg, bl, vol + vol_off, box, lane);
                    __builtin_amdgcn_s_waitcnt(0x0070);       // vmcnt(0) lgkmcnt(0): this wave's copies have landed
                    asm volatile("" ::: "memory");
                }
                // ---- pass 2: coordinates again, weights, gather; one 16-byte store per four voxels ---
                float o0 = 0.f, o1 = 0.f, o2 = 0.f;
                float* op = img + (img_off + obase + ox0);
                const int last = hg.out_len[2] - 1;
                XEntry xe;
                xentry_load(xt, min(ox0, last), xe);
#pragma unroll 1
                for (int k = 0; k < nx; ++k) {
                    const int ox = ox0 + k;
                    int st[3];
                    float fr[3];
                    XEntry xn;
                    xentry_load(xt, min(ox + 1, last), xn);       // (requested a whole voxel ahead)
                    const bool live = !entry_voxel<ORDER, AFFINE>(hg, hp, rw, qc, xe, ox, st, fr) && rw.vzy && ox <= last;
                    float w0[NT], w1[NT], w2[NT];
                    weights_from_frac<float, ORDER>(fr[0], w0);
                    weights_from_frac<float, ORDER>(fr[1], w1);
                    weights_from_frac<float, ORDER>(fr[2], w2);
                    // (a voxel that is not gathered reads the head of the box: no divergence)
                    const int rz = live ? st[0] - bl.b0[0] : 0, ry = live ? st[1] - bl.b0[1] : 0,
                              rx = live ? st[2] - bl.b0[2] : 0;
                    const float* bp = box + (rz * ps + ry * pitch + (rx & ~1));
                    float val = 0.f;
#ifdef EDHIP_EXPERIMENTS
                    if (ED_DBG(hg.dbg, 1))            // ablation: no gather
                        val = fr[0] + fr[1] + fr[2] + w0[1];
                    else
#endif
                    if (bl.any)
                        val = pitch == 16 ? wave_gather<ORDER, gather_bufs<ORDER, OCC>(), 16>(bp, ps, rx & 1, w0, w1, w2)
                                          : wave_gather<ORDER, gather_bufs<ORDER, OCC>(), 12>(bp, ps, rx & 1, w0, w1, w2);
                    val = live ? val : hg.cval;
                    xe = xn;
                    const int j = k & 3;       // (k is wave-uniform)
                    if (j == 0)
                        o0 = val;
                    else if (j == 1)
                        o1 = val;
                    else if (j == 2)
                        o2 = val;
                    else if (rw.vzy) {
                        // streaming stores: the wave fills 16-byte pieces of 64 rows; the rest of each
                        // 128-byte line follows from this same wave within the strip
                        float* o = op + (k - 3);
#ifdef EDHIP_EXPERIMENTS
                        if (ED_DBG(hg.dbg, 8) && val != -12345.678f)         // ablation: no stores
                            continue;
#endif
                        if (ox <= last) {
                            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                            f4u v4;
                            v4.x = o0;
                            v4.y = o1;
                            v4.z = o2;
                            v4.w = val;
                            __builtin_nontemporal_store(v4, reinterpret_cast<f4u*>(o));
                        } else {
                            if (ox - 3 <= last)
                                __builtin_nontemporal_store(o0, &o[0]);
                            if (ox - 2 <= last)
                                __builtin_nontemporal_store(o1, &o[1]);
                            if (ox - 1 <= last)
                                __builtin_nontemporal_store(o2, &o[2]);
                        }
                    }
                }
            }
        }
        if (beyond && hg.hint && lane == 0)
            atomicAdd(hg.hint, 1);
    }
    }
#ifdef EDHIP_EXPERIMENTS
    if (ED_DBG_PTR(hg.dbgbuf) && lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = hg.dbgbuf + (size_t)blockIdx.x * 4;
        d[0] = t_begin;
        d[1] = __builtin_readcyclecounter();
        d[2] = hwid;
        d[3] = xcc;
    }
#endif
}

// ================================================================================================
// K2: gradient.  Same walk; the box is an accumulator of fixed-point cells (ds_add_u32 sustains one
// wave-instruction per ~4 cycles on MI355X, ds_add_f32 one per ~190: profiles/r02_ubench_lds.txt)
// in a per-tile scale derived from the tile's sum of |dY|, flushed by its own wave with one float
// atomic per touched source element (deform.c:926-997; see deform_tile.hip for the no-overflow bound).
// ================================================================================================
__device__ __forceinline__ int round_half_up_i32(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));      // (int)floor(x + 0.5) in one instruction
    return r;
}

struct CellLayout {
    int b0[3], ext[3];
    int px, ps;           // cells per row / per plane
    bool any, fits;
};
__device__ __forceinline__ void grad_layout(const int (&lo)[3], const int (&hi)[3], int box_cap, CellLayout& cl)
{
    cl.any = hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        cl.b0[h] = lo[h];
        cl.ext[h] = cl.any ? hi[h] - lo[h] + 1 : 0;
    }
    // odd row pitch: the lanes of a wave sit on different (z, y) rows of the box at about the same x, and
    // an even pitch folds them onto few banks (tools/sim/conflicts_k2.py: 2.7 -> 1.9 passes per 16 lanes)
    cl.px = cl.ext[2] | 1;
    cl.ps = cl.ext[1] * cl.px;
    cl.fits = (unsigned)cl.ext[0] <= 1024u && (unsigned)cl.ext[1] <= 1024u && (unsigned)cl.ext[2] <= 1024u &&
              cl.ext[0] * cl.ps <= box_cap;
}

template <int ORDER, bool AFFINE, int OCC>
__global__ __launch_bounds__(64, OCC) void wave_grad_kernel(const HotGeom hg, const AxTab* __restrict__ xt,
                                                          const double* __restrict__ qtab,
                                                          const float* __restrict__ dy0, float* __restrict__ dx0)
{
    constexpr int NT = ORDER + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem);
    int* cells = reinterpret_cast<int*>(smem + kWaveHead);
    const int lane = threadIdx.x;
    WaveStrip sp;
    if (!wave_strip(hg, sp, blockIdx.x))
        return;
    // the accumulator cells start at zero and every flush leaves the cells it read at zero again
    for (int e = lane * 4; e < hg.box_cap; e += 256)
        *reinterpret_cast<int4*>(cells + e) = make_int4(0, 0, 0, 0);
    wave_prologue(hg, smem, lane);
    // lane -> (z, y): the 16 lanes that go through the LDS atomic unit together hold rows two apart
    // along z and y.  Neighbouring voxels share a window start wherever the deformation compresses,
    // and two lanes adding into one cell serialise the atomic (4.2 -> 6.0 cycles).
    int zz = 2 * (lane & 3) + ((lane >> 4) & 1);
    int yy = 2 * ((lane >> 2) & 3) + ((lane >> 5) & 1);
#ifdef EDHIP_EXPERIMENTS
    if (ED_DBG(hg.dbg, 16)) {          // experiment: lane = 8 z + y
        zz = lane >> 3;
        yy = lane & 7;
    }
#endif
    {
    const float* __restrict__ dy = dy0 + sp.sample * hg.img_bstride;
    float* __restrict__ dx = dx0 + sp.sample * hg.vol_bstride;

    RowWalk rw;
    rw.oz = sp.tz * kT + zz;
    rw.oy = sp.ty * kT + yy;
    rw.vzy = rw.oz < hg.out_len[0] && rw.oy < hg.out_len[1];
    rw.qrow = qtab + sp.sample * hg.q_bstride +
              ((long long)min(rw.oz, hg.out_len[0] - 1) * hg.out_len[1] + min(rw.oy, hg.out_len[1] - 1)) * (4 * hg.ncpx);
#pragma unroll
    for (int h = 0; h < 3; ++h)
        rw.Pzy[h] = AFFINE ? fma(hp->affine[h * 4 + 0], (double)rw.oz,
                                 fma(hp->affine[h * 4 + 1], (double)rw.oy, hp->affine[h * 4 + 3] + hp->offd[h]))
                           : 0.0;
    const int obase = rw.oz * hg.img_sz + rw.oy * hg.img_sy;
    QCols qc;
    {
        XEntry xe;
        xentry_load(xt, min(sp.tx0 * kT, hg.out_len[2] - 1), xe);
        qcols_load(qc, rw.qrow, xe.idx);
    }
    // |sum in a cell| <= max tap weight * sum over the tile of |dY|: this scale cannot overflow
    // (the 0.1 % margin covers the two roundings of the reciprocal and the product)
    constexpr float kC = (float)((2147483648.0 - 1024.0) /
                                 ((ORDER == 1 ? 1.0 : ORDER == 2 ? 0.4219 : ORDER == 3 ? 0.2963
                                   : ORDER == 4 ? 0.2150 : 0.1664) * 1.001));

    for (int ti = 0; ti < sp.ntile; ++ti) {
        const int tile_id = sp.sample * hg.ntiles + (sp.tz * hg.tiles[1] + sp.ty) * hg.tiles[2] + sp.tx0 + ti;
        const int ox_tile = (sp.tx0 + ti) * kT;
        // the tile's box: handed over by the forward call (EDHIP_FLAG_USE_BOXES) or pass 1
        bool given = hg.use_boxes != 0;
        int lo[3], hi[3];
        if (given) {
            const int* bx = hg.boxes + (size_t)tile_id * 8;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                lo[h] = uni(bx[h]);
                hi[h] = uni(bx[3 + h]);
            }
            // An EMPTY handed-over box (every voxel of the forward tile mapped to the constant) is stored with the
            // reduction's start values, INT_MAX / INT_MIN.  It may be stale -- under THIS call's grid the tile can have
            // live voxels, whose windows are tested against the box below: `start - INT_MAX` wraps for a window start of
            // -2 (orders 4 / 5 at the array's corner), the test then passed on all three axes and the voxel's taps went
            // to cells outside LDS, i.e. nowhere (found by tests/fuzz/fuzz_hot.py seed 501, round 6; there since round 3).
            // Canonical empty box instead: every live voxel fails the window test and scatters directly.
            if (!(hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2])) {
                lo[0] = lo[1] = lo[2] = 0;
                hi[0] = hi[1] = hi[2] = -1;
            }
        } else
            row_box<ORDER, AFFINE>(hg, hp, rw, qc, xt, ox_tile, kT, lo, hi);
        CellLayout cl;
        grad_layout(lo, hi, hg.box_cap, cl);
        int nsub = 1, nx = kT;
        CellLayout half0, half1;
        half0 = half1 = cl;
        // spill feedback: the tile would not fit the standard boxes (a whole tile that fits the large box
        // only: its halves are taken as three quarters of it)
        bool beyond = cl.fits && cl.ext[0] * cl.ps * 3 > hg.small_cap * 4;
        if (!cl.fits) {
            // two halves along x, each with a box of its own; nothing has been scattered yet, so a
            // tile whose halves do not fit either can still go to the general kernels as a whole
            given = false;
            row_box<ORDER, AFFINE>(hg, hp, rw, qc, xt, ox_tile, kT / 2, lo, hi);
            grad_layout(lo, hi, hg.box_cap, half0);
            row_box<ORDER, AFFINE>(hg, hp, rw, qc, xt, ox_tile + kT / 2, kT / 2, lo, hi);
            grad_layout(lo, hi, hg.box_cap, half1);
            beyond = !(half0.fits && half1.fits) || half0.ext[0] * half0.ps > hg.small_cap ||
                     half1.ext[0] * half1.ps > hg.small_cap;
            if (beyond && hg.hint && lane == 0)
                atomicAdd(hg.hint, 1);
            beyond = false;
            if (!(half0.fits && half1.fits)) {
                if (lane == 0) {
                    const int slot = atomicAdd(&hg.spill[0], 1);
                    hg.spill[1 + slot] = tile_id;
                }
                continue;
            }
            nsub = 2;
            nx = kT / 2;
        }
        if (beyond && hg.hint && lane == 0)
            atomicAdd(hg.hint, 1);
        for (int sub = 0; sub < nsub; ++sub) {
            if (nsub == 2)
                cl = sub ? half1 : half0;
            if (!cl.any && !given)
                continue;          // nothing to scatter (uniform): every voxel maps to the constant
            const int ox0 = ox_tile + sub * nx;
            const int px = cl.px, ps = cl.ps, ex = cl.ext[2];
            const bool interior = cl.b0[0] >= 0 && cl.b0[0] + cl.ext[0] <= hg.in_len[0] && cl.b0[1] >= 0 &&
                                  cl.b0[1] + cl.ext[1] <= hg.in_len[1] && cl.b0[2] >= 0 &&
                                  cl.b0[2] + cl.ext[2] <= hg.in_len[2];
            for (long long ss = 0; ss < hg.nsteps; ++ss) {
                long long vol_off = 0, img_off = 0;
                if (hg.nstep)
                    wave_step_offsets(hp, ss, vol_off, img_off);
                float* dst = dx + vol_off;
                // dY of the lane's row
                float g[kT];
                {
                    const float* gp = dy + (img_off + obase + ox0);
                    float gm = 0.f;
#pragma unroll
                    for (int k = 0; k < kT; ++k) {
                        g[k] = (k < nx && rw.vzy && ox0 + k < hg.out_len[2]) ? gp[k] : 0.f;
                        // inf / NaN gradients have no fixed-point scale: left out of the sum, scattered
                        // with float atomics below
                        gm += (__float_as_int(g[k]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(g[k]);
                    }
                    gm = wave_sum(gm);
                    const float scale = gm > 0.f ? fminf(kC * __frcp_rn(gm), 3.0e38f) : 0.f;
                    const float inv_scale = gm > 0.f ? __frcp_rn(scale) : 0.f;

                    // ---- pass 2: coordinates again, weights, scatter into the cells -------------------
                    const int last = hg.out_len[2] - 1;
                    XEntry xe;
                    xentry_load(xt, min(ox0, last), xe);
#pragma unroll 1
                    for (int k = 0; k < nx; ++k) {
                        float gv = g[0];
#pragma unroll
                        for (int j = 1; j < kT; ++j)
                            gv = k == j ? g[j] : gv;       // (k is wave-uniform)
                        const int ox = ox0 + k;
                        int st[3];
                        float fr[3];
                        XEntry xn;
                        xentry_load(xt, min(ox + 1, last), xn);       // (requested a whole voxel ahead)
                        const bool cst = entry_voxel<ORDER, AFFINE>(hg, hp, rw, qc, xe, ox, st, fr);
                        xe = xn;
                        if (gv == 0.f || cst)
                            continue;      // constant voxels contribute nothing (deform.c:928)
                        float w0[NT], w1[NT], w2[NT];
                        weights_from_frac<float, ORDER>(fr[0], w0);
                        weights_from_frac<float, ORDER>(fr[1], w1);
                        weights_from_frac<float, ORDER>(fr[2], w2);
                        const int rz = st[0] - cl.b0[0], ry = st[1] - cl.b0[1], rx = st[2] - cl.b0[2];
                        // boxes handed over by the forward call are a hint: a window outside goes the direct way
                        const bool outside = given && (rz < 0 || rz + ORDER >= cl.ext[0] || ry < 0 ||
                                                       ry + ORDER >= cl.ext[1] || rx < 0 || rx + ORDER >= cl.ext[2]);
                        if ((__float_as_int(gv) & 0x7f800000) == 0x7f800000 || outside) {
                            // inf / NaN gradient (no fixed-point scale), or a window outside a stale box:
                            // this voxel scatters its taps with float atomics straight to global memory
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
                        int* bp = cells + (rz * ps + ry * px + rx);
                        const float gs = gv * scale;
#pragma unroll
                        for (int l0 = 0; l0 < NT; ++l0) {
                            const float g0 = gs * w0[l0];
#pragma unroll
                            for (int l1 = 0; l1 < NT; ++l1) {
                                const float g1 = g0 * w1[l1];
                                int* rp = bp + (l0 * ps + l1 * px);
#pragma unroll
                                for (int l2 = 0; l2 < NT; ++l2)
                                    atomicAdd(reinterpret_cast<unsigned*>(rp + l2),
                                              (unsigned)round_half_up_i32(g1 * w2[l2]));
                            }
                        }
                    }
                    // ---- flush: one float atomic per touched source element (mirror-mapped at the edges,
                    //      deform.c:791-813); the wave's own DS operations are ordered, no barrier ----------
                    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0)
                    asm volatile("" ::: "memory");
                    {
                        constexpr int FU = 4;                  // rows in flight per lane
                        const int nrows = cl.ext[0] * cl.ext[1];
                        const float inv_by = 1.f / (float)cl.ext[1];
                        // 16 lanes per box row (4 rows per instruction) while the rows are that short
                        const int fl = ex <= 16 ? 16 : 64, fr_ = 64 / fl;
                        const int subl = lane & (fl - 1), rslot = ex <= 16 ? lane >> 4 : 0;
                        for (int xo = 0; xo < ex; xo += fl) {
                            const int xi = xo + subl;
                            const bool xin = xi < ex;
                            const int xs = interior ? cl.b0[2] + xi : mirror_i32(cl.b0[2] + min(xi, ex - 1), hg.in_len[2]);
                            for (int r0 = rslot; r0 < nrows; r0 += FU * fr_) {
                                int acc[FU];
#pragma unroll
                                for (int u = 0; u < FU; ++u) {
                                    const int r = r0 + u * fr_;
                                    // read and reset in one LDS operation (ds_wrxchg_rtn_b32)
                                    acc[u] = (xin && r < nrows)
                                                 ? __hip_atomic_exchange(&cells[r * px + xi], 0, __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_WORKGROUP)
                                                 : 0;
                                }
#pragma unroll
                                for (int u = 0; u < FU; ++u) {
                                    if (acc[u] != 0) {
                                        const int r = r0 + u * fr_;
                                        const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * cl.ext[1];
                                        int rowoff;
                                        if (interior)
                                            rowoff = (cl.b0[0] + zr) * hg.vol_sz + (cl.b0[1] + yr) * hg.vol_sy;
                                        else
                                            rowoff = mirror_i32(cl.b0[0] + zr, hg.in_len[0]) * hg.vol_sz +
                                                     mirror_i32(cl.b0[1] + yr, hg.in_len[1]) * hg.vol_sy;
                                        unsafeAtomicAdd(dst + (rowoff + xs), (float)acc[u] * inv_scale);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    }
}

// ================================================================================================
// Integer volumes (8- and 16-bit), spline orders 1-5, forward.  The reference runs every dtype through
// one double-precision loop (deform.c:863-887) and rounds half away from zero into the type
// (:292-306,906-919); integer results therefore have to be BIT-equal, and until now only the exact
// kernel (one thread per voxel, the reference's evaluation order: 3.1 ms for 256^3 int16, order 3)
// served them.  Here: the same walk as the float kernel, the box staged as float32 (exact for these
// types), coordinates from the per-call tables, weights and taps in fp64.  That value is within
// ~2e-5 of the reference's (the fast coordinate is within ~1e-11 of the reference's, |v| < 2^16), so
// every voxel rounds identically EXCEPT where the value lies within 1e-4 of a rounding tie (x.5), or
// where the coordinate sits on a decision boundary of the boundary map (an integer once it is at or
// beyond the array's ends: the constant test, the fold points).  Those voxels -- a few per ten
// thousand -- are listed and re-evaluated by the exact kernel (launch_deform_exact_list), exactly like
// the near-tie voxels of the order-0 label kernel (deform_tile.hip).
// ================================================================================================
constexpr double kIntTieBand = 1e-4;
constexpr double kCoordEps = 1e-6;

// coordinates with the fraction in fp64; `near` is set when the voxel has to be re-evaluated exactly
template <int ORDER, bool AFFINE>
__device__ __forceinline__ bool int_coords(const HotGeom& hg, const HotParams* hp, const QCols& qc,
                                           const double (&tw)[4], const int (&b)[3], const double (&P)[3],
                                           int* start, double* frac, bool& near)
{
    double d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        d[c] = tw[0] * qc.v[0][c];
#pragma unroll
    for (int l = 1; l < 4; ++l)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            d[c] = fma(tw[l], qc.v[l][c], d[c]);
    int ci[3];
    bool inr[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        inr[h] = coord_axis_fast<ORDER, double>(AFFINE ? P[h] + d[h] : d[h], AFFINE ? 0 : b[h], hg.in_len[h],
                                                ci[h], frac[h]);
        // the outermost cells take the slow region too: the array's ends are decision boundaries
        constexpr int lo = (ORDER & 1) ? 0 : 1;
        const int hi = (ORDER & 1) ? hg.in_len[h] - 2 : hg.in_len[h] - 2;
        inr[h] = inr[h] && ci[h] > lo && ci[h] < hi;
    }
    bool cst = false;
    near = false;
    if (!(inr[0] && inr[1] && inr[2])) {
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            if (!inr[h]) {
                const double c = AFFINE ? P[h] + d[h] : (double)b[h] + d[h];
                const double fr = c - floor(c);
                const bool outside = !(c >= kCoordEps && c <= (double)(hg.in_len[h] - 1) - kCoordEps);
                near = near || (outside && (fr < kCoordEps || fr > 1.0 - kCoordEps)) || !(c == c);
                cst = coord_axis_mapped<ORDER, double>(c, hg.in_len[h], hg.mode, hp->period[h], hp->inv_period[h],
                                                       ci[h], frac[h]) || cst;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 3; ++h)
        start[h] = cst ? 0 : ci[h] - ORDER / 2;
    return cst;
}

// (ORDER + 1)^3 taps in fp64 from the float32 box (column-first contraction, see wave_gather for the
// pair / parity scheme)
// KIND: what a 4-byte cell of the box holds -- 0: the value as float32 (8- / 16-bit volumes, exact), 1: the bits
// of an int32, 2: of a uint32 (32-bit volumes do not fit a float32 mantissa; v_cvt_f64_i32 costs what
// v_cvt_f64_f32 does)
template <int KIND>
__device__ __forceinline__ double int_cell(float v)
{
    return KIND == 0 ? (double)v : (KIND == 1 ? (double)__float_as_int(v) : (double)__float_as_uint(v));
}

template <int ORDER, int PITCH, int KIND>
__device__ __forceinline__ double int_gather(const float* bp, int ps, bool par, const double (&w0)[ORDER + 1],
                                             const double (&w1)[ORDER + 1], const double (&w2)[ORDER + 1])
{
    constexpr int NT = ORDER + 1;
    constexpr int NP = (NT + 2) / 2;
    constexpr int NC = NT + 1;
    double S[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j)
        S[j] = 0.0;
#pragma unroll
    for (int l0 = 0; l0 < NT; ++l0) {
        const float* pp = bp + l0 * ps;
        float v[NT][2 * NP];
#pragma unroll
        for (int l1 = 0; l1 < NT; ++l1) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float2 pr = *reinterpret_cast<const float2*>(pp + l1 * PITCH + 2 * p);
                ED_NO_DS_MERGE();
                v[l1][2 * p] = pr.x;
                v[l1][2 * p + 1] = pr.y;
            }
            if (2 * NP > NC)
                asm volatile("" ::"v"(v[l1][2 * NP - 1]));
        }
        double c[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j)
            c[j] = 0.0;
#pragma unroll
        for (int l1 = 0; l1 < NT; ++l1)
#pragma unroll
            for (int j = 0; j < NC; ++j)
                c[j] = fma(w1[l1], int_cell<KIND>(v[l1][j]), c[j]);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            S[j] = fma(w0[l0], c[j], S[j]);
            asm volatile("" : "+v"(S[j]));
        }
    }
    S[0] = par ? 0.0 : S[0];
    S[NT] = par ? S[NT] : 0.0;
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const double wl = j > 0 ? w2[j - 1] : 0.0, wr = j < NT ? w2[j] : 0.0;
        a = fma(par ? wl : wr, S[j], a);
    }
    return a;
}

// the reference's store rule (deform.c:292-306): round half away from zero, clamp, C cast
template <typename TIN>
__device__ __forceinline__ TIN int_store_value(double t)
{
    constexpr bool is_signed = (TIN)(-1) < (TIN)0;
    constexpr double lo = is_signed ? -(double)(1ull << (8 * sizeof(TIN) - 1)) : 0.0;
    constexpr double hi = is_signed ? (double)((1ull << (8 * sizeof(TIN) - 1)) - 1) : (double)((1ull << (8 * sizeof(TIN))) - 1);
    t = t > 0 ? t + 0.5 : (is_signed ? t - 0.5 : 0.0);
    t = t > hi ? hi : t;
    t = t < lo ? lo : t;
    if (sizeof(TIN) == 4)
        return (TIN)(long long)t;
    return (TIN)(int)t;          // (NaN: the tie test below sends the voxel to the exact kernel)
}

template <int ORDER, bool AFFINE, typename TIN>
__global__ __launch_bounds__(64, 3) void wave_int_fwd_kernel(const HotGeom hg, const AxTab* __restrict__ xt,
                                                             const double* __restrict__ qtab,
                                                             const TIN* __restrict__ vol0, TIN* __restrict__ img0)
{
    constexpr int NT = ORDER + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem);
    float* box = reinterpret_cast<float*>(smem + kWaveHead);
    const int lane = threadIdx.x;
    WaveStrip sp;
    if (!wave_strip(hg, sp, blockIdx.x))
        return;
    wave_prologue(hg, smem, lane);
    const int yy = lane >> 3, zz = lane & 7;
    const TIN* __restrict__ vol = vol0;
    TIN* __restrict__ img = img0;

    RowWalk rw;
    rw.oz = sp.tz * kT + zz;
    rw.oy = sp.ty * kT + yy;
    rw.vzy = rw.oz < hg.out_len[0] && rw.oy < hg.out_len[1];
    rw.qrow = qtab + ((long long)min(rw.oz, hg.out_len[0] - 1) * hg.out_len[1] + min(rw.oy, hg.out_len[1] - 1)) * (4 * hg.ncpx);
#pragma unroll
    for (int h = 0; h < 3; ++h)
        rw.Pzy[h] = AFFINE ? fma(hp->affine[h * 4 + 0], (double)rw.oz,
                                 fma(hp->affine[h * 4 + 1], (double)rw.oy, hp->affine[h * 4 + 3] + hp->offd[h]))
                           : 0.0;
    const int obase = rw.oz * hg.img_sz + rw.oy * hg.img_sy;
    const int vox_row = (rw.oz * hg.out_len[1] + rw.oy) * hg.out_len[2];
    QCols qc;
    {
        XEntry xe;
        xentry_load(xt, min(sp.tx0 * kT, hg.out_len[2] - 1), xe);
        qcols_load(qc, rw.qrow, xe.idx);
    }
    const int last = hg.out_len[2] - 1;
    const TIN cst_value = int_store_value<TIN>(hg.cvald);
    constexpr int KIND = sizeof(TIN) < 4 ? 0 : (((TIN)(-1) < (TIN)0) ? 1 : 2);
    auto encode = [](TIN x) -> float {
        return KIND == 0 ? (float)x : (KIND == 1 ? __int_as_float((int)x) : __uint_as_float((unsigned)x));
    };
    // 32-bit volumes: the band around a rounding tie grows with the magnitude of the data (the fast and
    // the exact coordinates differ by ~1e-13 relative, and a tap sum of 2e9 carries that as 1e-3): 1e-9 of
    // the largest |sample| of the box -- the same margin the 1e-4 leaves a full-range int16 volume
    float boxmax = 0.0f;

    for (int ti = 0; ti < sp.ntile; ++ti) {
        const int ox_tile = (sp.tx0 + ti) * kT;
        // whole tile; two halves along x if its box does not fit; taps straight from global memory if
        // the halves do not fit either (mode 2)
        int nsub = 1, nx = kT;
        for (int sub = 0; sub < nsub; ++sub) {
            const int ox0 = ox_tile + sub * nx;
            int lo[3], hi[3];
            row_box<ORDER, AFFINE>(hg, hp, rw, qc, xt, ox0, nx, lo, hi);
            BoxLayout bl;
            fwd_layout<ORDER>(lo, hi, hg.box_cap, nsub == 2, bl);
            bool direct = false;
            if (bl.any && !bl.fits) {
                if (nsub == 1) {
                    nsub = 2;
                    nx = kT / 2;
                    sub = -1;
                    continue;
                }
                direct = true;
            }
            const int ps = bl.ps, pitch = bl.pitch;
            for (long long ss = 0; ss < hg.nsteps; ++ss) {
                long long vol_off = 0, img_off = 0;
                if (hg.nstep)
                    wave_step_offsets(hp, ss, vol_off, img_off);
                const TIN* __restrict__ src = vol + vol_off;
                if (bl.any && !direct) {
                    // stage the box as float32, every index through the mirror map (deform.c:791-813)
                    const int sx = lane & 15;
                    const int nrows = bl.ext[0] * bl.ext[1];
                    const float inv_by = 1.0f / (float)bl.ext[1];
                    const bool interior = bl.b0[0] >= 0 && bl.b0[0] + bl.ext[0] <= hg.in_len[0] && bl.b0[1] >= 0 &&
                                          bl.b0[1] + bl.ext[1] <= hg.in_len[1] && bl.b0[2] >= 0 &&
                                          bl.b0[2] + bl.ext[2] <= hg.in_len[2];
                    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): the previous gather's reads
                    float vmax = 0.0f;
                    if (interior) {
                        // four rows per wave instruction, 16 lanes along x; all loads of the box in flight
                        const TIN* p0 = src + ((bl.b0[0] * hg.vol_sz + bl.b0[1] * hg.vol_sy) + bl.b0[2] + sx);
#pragma unroll 4
                        for (int r = lane >> 4; r < nrows; r += 4) {
                            const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * bl.ext[1];
                            if (sx < bl.ext[2]) {
                                const TIN x = p0[zr * hg.vol_sz + yr * hg.vol_sy];
                                box[zr * ps + yr * pitch + sx] = encode(x);
                                if (KIND)
                                    vmax = fmaxf(vmax, fabsf((float)x));
                            }
                        }
                    } else {
                        const int xs = mirror_i32(bl.b0[2] + min(sx, bl.ext[2] - 1), hg.in_len[2]);
                        for (int r = lane >> 4; r < nrows; r += 4) {
                            const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * bl.ext[1];
                            const int zs = mirror_i32(bl.b0[0] + zr, hg.in_len[0]);
                            const int ys = mirror_i32(bl.b0[1] + yr, hg.in_len[1]);
                            if (sx < bl.ext[2]) {
                                const TIN x = src[zs * hg.vol_sz + ys * hg.vol_sy + xs];
                                box[zr * ps + yr * pitch + sx] = encode(x);
                                if (KIND)
                                    vmax = fmaxf(vmax, fabsf((float)x));
                            }
                        }
                    }
                    if (KIND) {
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1)
                            vmax = fmaxf(vmax, __shfl_xor(vmax, m));
                        boxmax = vmax;
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    asm volatile("" ::: "memory");
                }
                TIN* op = img + (img_off + obase);
                // the lane's voxels of this (sub-)tile are packed and stored together: 8 or 16 contiguous
                // bytes per lane instead of one 1- / 2-byte store per voxel
                unsigned long long pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0;
                constexpr int kBits = 8 * (int)sizeof(TIN);
                constexpr int kPer64 = 64 / kBits;             // voxels per 64-bit word: 8, 4 or 2
                XEntry xe;
                xentry_load(xt, min(ox0, last), xe);
#pragma unroll 1
                for (int k = 0; k < nx; ++k) {
                    const int ox = ox0 + k;
                    int st[3];
                    double fr[3];
                    bool near;
                    XEntry xn;
                    xentry_load(xt, min(ox + 1, last), xn);
                    const int b[3] = {rw.oz + hg.off[0], rw.oy + hg.off[1], ox + hg.off[2]};
                    double P[3] = {0.0, 0.0, 0.0};
                    if (xe.idx[0] != qc.idx[0] || xe.idx[1] != qc.idx[1] || xe.idx[2] != qc.idx[2] ||
                        xe.idx[3] != qc.idx[3])
                        qcols_load(qc, rw.qrow, xe.idx);
                    if (AFFINE) {
#pragma unroll
                        for (int h = 0; h < 3; ++h)
                            P[h] = fma(hp->affine[h * 4 + 2], (double)ox, rw.Pzy[h]);
                    }
                    const bool cst = int_coords<ORDER, AFFINE>(hg, hp, qc, xe.w, b, P, st, fr, near);
                    xe = xn;
                    const bool inside = rw.vzy && ox <= last;
                    const bool live = inside && !cst;
                    double w0[NT], w1[NT], w2[NT];
                    weights_from_frac<double, ORDER>(fr[0], w0);
                    weights_from_frac<double, ORDER>(fr[1], w1);
                    weights_from_frac<double, ORDER>(fr[2], w2);
                    double t = 0.0;
                    double band = kIntTieBand;
                    if (direct) {
                        float dmax = 0.0f;
                        if (live) {
                            // taps straight from global memory, mirror-mapped (rolled: rare)
#pragma unroll 1
                            for (int l0 = 0; l0 < NT; ++l0) {
                                const int zs = mirror_i32(st[0] + l0, hg.in_len[0]);
                                double a1 = 0.0;
#pragma unroll 1
                                for (int l1 = 0; l1 < NT; ++l1) {
                                    const int ys = mirror_i32(st[1] + l1, hg.in_len[1]);
                                    double a2 = 0.0;
#pragma unroll
                                    for (int l2 = 0; l2 < NT; ++l2) {
                                        const TIN x = src[zs * hg.vol_sz + ys * hg.vol_sy + mirror_i32(st[2] + l2, hg.in_len[2])];
                                        a2 = fma(w2[l2], (double)x, a2);
                                        if (KIND)
                                            dmax = fmaxf(dmax, fabsf((float)x));
                                    }
                                    double wy = w1[0];
#pragma unroll
                                    for (int l = 1; l < NT; ++l)
                                        wy = l1 == l ? w1[l] : wy;
                                    a1 = fma(wy, a2, a1);
                                }
                                double wz = w0[0];
#pragma unroll
                                for (int l = 1; l < NT; ++l)
                                    wz = l0 == l ? w0[l] : wz;
                                t = fma(wz, a1, t);
                            }
                        }
                        if (KIND)
                            band = kIntTieBand + 1e-9 * (double)dmax;
                    } else if (bl.any) {
                        if (KIND)
                            band = kIntTieBand + 1e-9 * (double)boxmax;
                        const int rz = live ? st[0] - bl.b0[0] : 0, ry = live ? st[1] - bl.b0[1] : 0,
                                  rx = live ? st[2] - bl.b0[2] : 0;
                        const float* bp = box + (rz * ps + ry * pitch + (rx & ~1));
                        t = pitch == 16 ? int_gather<ORDER, 16, KIND>(bp, ps, rx & 1, w0, w1, w2)
                                        : int_gather<ORDER, 12, KIND>(bp, ps, rx & 1, w0, w1, w2);
                    }
                    // value within the band of a rounding tie (x.5), or not a number: exact kernel
                    const double at = fabs(t);
                    const double ft = at - floor(at);
                    const bool tie = live && (!(fabs(ft - 0.5) >= band) || !(t == t));
                    if (inside) {
                        if ((near || tie) && hg.tie_list) {
                            const int slot = atomicAdd(&hg.tie_list[0], 1);
                            if (slot < hg.tie_cap)
                                hg.tie_list[1 + slot] = vox_row + ox;
                        }
                    }
                    {
                        typedef typename std::make_unsigned<TIN>::type UT;
                        const unsigned long long bits = (unsigned long long)(UT)(live ? int_store_value<TIN>(t) : cst_value);
                        const int word = k / kPer64, sh = kBits * (k % kPer64);        // (k is wave-uniform)
                        if (word == 0)
                            pk0 |= bits << sh;
                        else if (word == 1)
                            pk1 |= bits << sh;
                        else if (word == 2)
                            pk2 |= bits << sh;
                        else
                            pk3 |= bits << sh;
                    }
                }
                if (rw.vzy) {
                    TIN* o = op + ox0;
                    const int nbytes = nx * (int)sizeof(TIN);          // 4, 8, 16 or 32
                    const bool whole = ox0 + nx - 1 <= last && (((size_t)o) & (nbytes >= 8 ? 7 : 3)) == 0;
                    if (whole) {
                        unsigned long long* o8 = reinterpret_cast<unsigned long long*>(o);
                        if (nbytes == 4)
                            __builtin_nontemporal_store((unsigned)pk0, reinterpret_cast<unsigned*>(o));
                        else
                            __builtin_nontemporal_store(pk0, o8);
                        if (nbytes >= 16)
                            __builtin_nontemporal_store(pk1, o8 + 1);
                        if (nbytes == 32) {
                            __builtin_nontemporal_store(pk2, o8 + 2);
                            __builtin_nontemporal_store(pk3, o8 + 3);
                        }
                    } else {
                        for (int k = 0; k < nx; ++k)
                            if (ox0 + k <= last) {
                                const int word = k / kPer64;
                                const unsigned long long w = word == 0 ? pk0 : (word == 1 ? pk1 : (word == 2 ? pk2 : pk3));
                                o[k] = (TIN)(w >> (kBits * (k % kPer64)));
                            }
                    }
                }
            }
        }
    }
}

template <int ORDER, typename TIN>
hipError_t launch_wave_int_t(const HotGeom& hg, const void* vol, void* img, unsigned nblk, size_t lds, hipStream_t stream)
{
    if (hg.has_affine)
        hipLaunchKernelGGL((wave_int_fwd_kernel<ORDER, true, TIN>), dim3(nblk), dim3(64), lds, stream, hg, hg.xt, hg.q,
                           (const TIN*)vol, (TIN*)img);
    else
        hipLaunchKernelGGL((wave_int_fwd_kernel<ORDER, false, TIN>), dim3(nblk), dim3(64), lds, stream, hg, hg.xt, hg.q,
                           (const TIN*)vol, (TIN*)img);
    return hipGetLastError();
}
template <int ORDER>
hipError_t launch_wave_int_o(const HotGeom& hg, int dtype, const void* vol, void* img, unsigned nblk, size_t lds,
                             hipStream_t stream)
{
    switch (dtype) {
    case EDHIP_U8: return launch_wave_int_t<ORDER, uint8_t>(hg, vol, img, nblk, lds, stream);
    case EDHIP_I8: return launch_wave_int_t<ORDER, int8_t>(hg, vol, img, nblk, lds, stream);
    case EDHIP_U16: return launch_wave_int_t<ORDER, uint16_t>(hg, vol, img, nblk, lds, stream);
    case EDHIP_I16: return launch_wave_int_t<ORDER, int16_t>(hg, vol, img, nblk, lds, stream);
    case EDHIP_U32: return launch_wave_int_t<ORDER, uint32_t>(hg, vol, img, nblk, lds, stream);
    case EDHIP_I32: return launch_wave_int_t<ORDER, int32_t>(hg, vol, img, nblk, lds, stream);
    default: return hipErrorNotSupported;
    }
}

// bytes of LDS per wave (one workgroup = one wave): 160 KiB / 12 resp. / 16 in the hardware's
// allocation granule
constexpr int kWaveLdsBytes3 = 12800, kWaveLdsBytes4 = 10240;

template <int ORDER, int OCC>
hipError_t launch_wave_occ(const HotGeom& hg, bool gradient, unsigned nblk, size_t lds, hipStream_t stream)
{
    if (gradient) {
        if (hg.has_affine)
            hipLaunchKernelGGL((wave_grad_kernel<ORDER, true, OCC>), dim3(nblk), dim3(64), lds, stream, hg, hg.xt, hg.q,
                               hg.img_r, hg.vol_w);
        else
            hipLaunchKernelGGL((wave_grad_kernel<ORDER, false, OCC>), dim3(nblk), dim3(64), lds, stream, hg, hg.xt, hg.q,
                               hg.img_r, hg.vol_w);
        return hipGetLastError();
    }
    if (hg.has_affine)
        hipLaunchKernelGGL((wave_fwd_kernel<ORDER, true, OCC>), dim3(nblk), dim3(64), lds, stream, hg, hg.xt, hg.q,
                           hg.vol_r, hg.img_w);
    else
        hipLaunchKernelGGL((wave_fwd_kernel<ORDER, false, OCC>), dim3(nblk), dim3(64), lds, stream, hg, hg.xt, hg.q,
                           hg.vol_r, hg.img_w);
    return hipGetLastError();
}

template <int ORDER>
hipError_t launch_wave_order(const HotGeom& hg, bool gradient, unsigned nblk, size_t lds, int occ, hipStream_t stream)
{
#ifdef EDHIP_EXPERIMENTS
    if (occ == 4)
        return launch_wave_occ<ORDER, 4>(hg, gradient, nblk, lds, stream);
#endif
    (void)occ;
    return launch_wave_occ<ORDER, 3>(hg, gradient, nblk, lds, stream);
}

}  // namespace

// LDS per wave and the box capacity (floats / cells) that leaves
size_t wave_lds_bytes(bool gradient, int occ, int* box_cap)
{
    (void)gradient;
    const int bytes = occ == 4 ? kWaveLdsBytes4 : kWaveLdsBytes3;
    *box_cap = (bytes - kWaveHead) / 4;
    return bytes;
}

hipError_t launch_wave_int(const HotGeom& hg, int order, int dtype, const void* vol, void* img, unsigned nblk,
                           size_t lds, hipStream_t stream)
{
    switch (order) {
    case 1: return launch_wave_int_o<1>(hg, dtype, vol, img, nblk, lds, stream);
    case 2: return launch_wave_int_o<2>(hg, dtype, vol, img, nblk, lds, stream);
    case 3: return launch_wave_int_o<3>(hg, dtype, vol, img, nblk, lds, stream);
    case 4: return launch_wave_int_o<4>(hg, dtype, vol, img, nblk, lds, stream);
    case 5: return launch_wave_int_o<5>(hg, dtype, vol, img, nblk, lds, stream);
    default: return hipErrorNotSupported;
    }
}

hipError_t launch_wave_level1(const HotGeom& hg, int order, bool gradient, unsigned nblk, size_t lds, int occ,
                              hipStream_t stream)
{
    switch (order) {
    case 1: return launch_wave_order<1>(hg, gradient, nblk, lds, occ, stream);
    case 2: return launch_wave_order<2>(hg, gradient, nblk, lds, occ, stream);
    case 3: return launch_wave_order<3>(hg, gradient, nblk, lds, occ, stream);
    case 4: return launch_wave_order<4>(hg, gradient, nblk, lds, occ, stream);
    case 5: return launch_wave_order<5>(hg, gradient, nblk, lds, occ, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace tile
}  // namespace ed
