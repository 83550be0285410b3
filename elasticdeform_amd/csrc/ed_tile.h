// ed_tile.h -- definitions shared by the LDS-tiled deform kernels: deform_tile.hip (general tile
// kernels, tables, spill levels, label kernel) and deform_hot.hip (the float32 benchmark kernels).
#pragma once

#include "ed_device.h"
#include "ed_params.h"
#include "ed_workspace.h"

namespace ed {
namespace tile {

constexpr int kT = 8;                 // tile edge
constexpr int kStrip = 8;             // tiles per strip (along x)
constexpr int kBlock = 256;

struct AxTab {
    double w[4];
    int idx[4];
};

// 16 bytes at 4-byte alignment: compiles to one global_load_dwordx4 (unaligned access is on)
struct __attribute__((packed, aligned(4))) F4u {
    float x, y, z, w;
};
static_assert(sizeof(AxTab) == 48, "AxTab layout");

// Compiler fence between two LDS reads: stops the backend's load/store optimizer from fusing
// neighbouring ds_read_b64 into ds_read2_b64.  Emits no instruction; s_waitcnt insertion is unaffected.
#define ED_NO_DS_MERGE() asm volatile("" ::: "memory")

// LDS carve (bytes)
constexpr int kOffTabX = 0;                               // [kStrip][8] AxTab
constexpr int kOffRed = kOffTabX + kStrip * kT * 48;      // int[3][8]: lo[3], hi[3], -, - (triple-buffered)
constexpr int kOffSum = kOffRed + 96;                     // float[2][4]: per-wave sum |dY| (K2)
constexpr int kOffHot = kOffSum + 32;                     // HotParams (uniform values kept out of SGPRs)
constexpr int kOffQ = kOffHot + 416;
static_assert(kOffQ % 16 == 0, "LDS carve alignment");

// Uniform per-call values the per-voxel code needs.  Kept in LDS and fetched with broadcast reads:
// as kernel arguments they would be hoisted into ~100 scalar registers, live across the whole tile
// loop, and the spill reloads (v_readlane) were a third of the kernel's VALU work.
struct HotParams {
    double offd[3];        // crop offset per axis
    double last[3];        // I_k - 1
    double period[3];      // boundary-map period for the input's mode, and its reciprocal
    double inv_period[3];
    double affine[12];     // inverse map, 3 x 4
    long long step_len[8];
    long long in_step_stride[8];
    long long out_step_stride[8];
    int nstep;
    int pad_;
    double slack[3];       // K1: margin of the sampled tile boxes (HotGeom::slack), per axis
};
static_assert(sizeof(HotParams) <= 416, "HotParams must fit its LDS slot");

__device__ __forceinline__ int mirror_i32(int idx, int len)
{
    if ((unsigned)idx < (unsigned)len)
        return idx;                       // in range: the common case
    if (len <= 1)
        return 0;
    if (idx < 0 && idx > -len)
        return -idx;                      // one reflection at the low end
    if (idx >= len && idx <= 2 * len - 2)
        return 2 * len - 2 - idx;         // one reflection at the high end
    const int period = 2 * len - 2;
    if (idx < 0) {
        idx = period * (-idx / period) + idx;
        idx = idx <= 1 - len ? idx + period : -idx;
    } else {
        idx -= period * (idx / period);
        if (idx >= len)
            idx = period - idx;
    }
    return idx;
}

// Wave-wide min / max / sum without touching LDS: four DPP steps fold each row of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row
// results meet in scalar registers.  All 64 lanes must be active.
#define ED_DPP_STEP(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ int wave_min(int v)
{
    v = min(v, ED_DPP_STEP(v, 0xB1));
    v = min(v, ED_DPP_STEP(v, 0x4E));
    v = min(v, ED_DPP_STEP(v, 0x141));
    v = min(v, ED_DPP_STEP(v, 0x140));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v)
{
    v = max(v, ED_DPP_STEP(v, 0xB1));
    v = max(v, ED_DPP_STEP(v, 0x4E));
    v = max(v, ED_DPP_STEP(v, 0x141));
    v = max(v, ED_DPP_STEP(v, 0x140));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ float wave_sum(float f)
{
    f += __int_as_float(ED_DPP_STEP(__float_as_int(f), 0xB1));
    f += __int_as_float(ED_DPP_STEP(__float_as_int(f), 0x4E));
    f += __int_as_float(ED_DPP_STEP(__float_as_int(f), 0x141));
    f += __int_as_float(ED_DPP_STEP(__float_as_int(f), 0x140));
    const int v = __float_as_int(f);
    return (__int_as_float(__builtin_amdgcn_readlane(v, 0)) +
            __int_as_float(__builtin_amdgcn_readlane(v, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(v, 32)) +
            __int_as_float(__builtin_amdgcn_readlane(v, 48)));
}

// Boundary map of a real coordinate for the fast kernels: the same piecewise map as deform.c:47-128
// (legacy SciPy semantics, same branch structure) with the trunc-divisions `(npy_intp)(c / period)`
// replaced by floor(c * (1 / period)) -- equal as real functions (every argument is positive
// there), different only in the last ulp of the product, which the fast path does not promise.
__device__ __forceinline__ double map_coordinate_fast(double c, int len, int mode, double period,
                                                      double inv_period)
{
    const double last = (double)(len - 1);
    if (c < 0) {
        switch (mode) {
        case EDHIP_MODE_MIRROR:
            if (len <= 1) {
                c = 0;
            } else {
                c = period * floor(-c * inv_period) + c;
                c = c <= -last ? c + period : -c;
            }
            break;
        case EDHIP_MODE_REFLECT:
            if (len <= 1) {
                c = 0;
            } else {
                if (c < -period)
                    c = period * floor(-c * inv_period) + c;
                c = c < (double)(-len) ? c + period : -c - 1;
            }
            break;
        case EDHIP_MODE_WRAP:
            if (len <= 1)
                c = 0;
            else
                c += period * (floor(-c * inv_period) + 1);
            break;
        case EDHIP_MODE_NEAREST: c = 0; break;
        default: c = -1; break;   // constant
        }
    } else if (c > last) {
        switch (mode) {
        case EDHIP_MODE_MIRROR:
            if (len <= 1) {
                c = 0;
            } else {
                c -= period * floor(c * inv_period);
                if (c >= (double)len)
                    c = period - c;
            }
            break;
        case EDHIP_MODE_REFLECT:
            if (len <= 1) {
                c = 0;
            } else {
                c -= period * floor(c * inv_period);
                if (c >= (double)len)
                    c = period - c - 1;
            }
            break;
        case EDHIP_MODE_WRAP:
            if (len <= 1)
                c = 0;
            else
                c -= period * floor(c * inv_period);
            break;
        case EDHIP_MODE_NEAREST: c = last; break;
        default: c = -1; break;   // constant
        }
    }
    return c;
}

// B-spline basis weights from the fractional offset, in the data's own width.  Same closed forms
// as deform.c:160-268 (last weight = 1 - sum of the others) with the divisions folded into
// constants.  x is c - floor(c) (odd orders) or c - floor(c + 0.5) (even orders).
// Every product-sum is written out as the fused multiply-add it is meant to be and the compiler's own
// contraction is switched off: the function is inlined into a dozen kernels, and left to itself the
// backend fused `0.75 - x * x` in some of them and not in others -- a voxel then got weights an ulp
// apart depending on the level that served its tile (order 2: 6 % of the voxels).
__device__ __forceinline__ float wt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double wt_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T, int ORDER>
__device__ __forceinline__ void weights_from_frac(T x, T* w)
{
#pragma clang fp contract(off)
    const T z = (T)1 - x;
    if (ORDER == 0) {
        w[0] = (T)1;
        return;
    }
    if (ORDER == 1) {
        w[0] = z;
    } else if (ORDER == 2) {
        w[1] = wt_fma(-x, x, (T)0.75);
        const T y = (T)0.5 - x;
        w[0] = ((T)0.5 * y) * y;
    } else if (ORDER == 3) {
        // all four cubic pieces in closed form (cheaper than "last = 1 - sum"; same polynomials)
        const T x2 = x * x, z2 = z * z;
        w[0] = (z2 * z) * (T)(1.0 / 6.0);
        w[3] = (x2 * x) * (T)(1.0 / 6.0);
        w[1] = wt_fma(x2, wt_fma(x, (T)0.5, (T)-1), (T)(2.0 / 3.0));
        w[2] = wt_fma(z2, wt_fma(z, (T)0.5, (T)-1), (T)(2.0 / 3.0));
        return;
    } else if (ORDER == 4) {
        T t = x * x;
        w[2] = wt_fma(t, wt_fma(t, (T)0.25, (T)-0.625), (T)(115.0 / 192.0));
        T y = (T)1 + x;
        w[1] = wt_fma(y, wt_fma(y, wt_fma(y * ((T)5 - y), (T)(1.0 / 6.0), (T)-1.25), (T)(5.0 / 24.0)), (T)(55.0 / 96.0));
        w[3] = wt_fma(z, wt_fma(z, wt_fma(z * ((T)5 - z), (T)(1.0 / 6.0), (T)-1.25), (T)(5.0 / 24.0)), (T)(55.0 / 96.0));
        y = (T)0.5 - x;
        t = y * y;
        w[0] = (t * t) * (T)(1.0 / 24.0);
    } else {
        T t = x * x;
        w[2] = wt_fma(t, wt_fma(t, wt_fma(-x, (T)(1.0 / 12.0), (T)0.25), (T)-0.5), (T)0.55);
        t = z * z;
        w[3] = wt_fma(t, wt_fma(t, wt_fma(-z, (T)(1.0 / 12.0), (T)0.25), (T)-0.5), (T)0.55);
        T y = x + (T)1;
        w[1] = wt_fma(y, wt_fma(y, wt_fma(y, wt_fma(y, wt_fma(y, (T)(1.0 / 24.0), (T)-0.375), (T)1.25), (T)-1.75), (T)0.625),
                      (T)0.425);
        const T zz = z + (T)1;
        w[4] = wt_fma(zz, wt_fma(zz, wt_fma(zz, wt_fma(zz, wt_fma(zz, (T)(1.0 / 24.0), (T)-0.375), (T)1.25), (T)-1.75), (T)0.625),
                      (T)0.425);
        y = (T)1 - x;
        t = y * y;
        w[0] = ((y * t) * t) * (T)(1.0 / 120.0);
    }
    T last = (T)1;
#pragma unroll
    for (int i = 0; i < ORDER; ++i)
        last -= w[i];
    w[ORDER] = last;
}

// ---- one axis of the source coordinate (deform.c:771-788) on the fast path -------------------------
// Every tile kernel (hot, general, direct) splits a coordinate into window start + fraction with
// these three functions, so a voxel gets bit-identical (start, frac) whichever kernel serves its
// tile -- the crop identity full[crop] == cropped (README.md:113) holds bit for bit across the spill
// levels.  Without an affine map the integer part (output index + crop offset, `b`) never meets the
// fp64 sum: floor(b + d) = b + floor(d).  Return value: the window start needs no boundary map
// (odd orders 0 <= floor(c) <= I-2; even orders 1 <= floor(c + 0.5) <= I-2).
template <int ORDER, typename T>
__device__ __forceinline__ bool coord_axis_fast(double d, int b, int in_len, int& ci, T& frac)
{
    const double fl = floor((ORDER & 1) ? d : d + 0.5);
    ci = (int)fl + b;
    frac = (T)(d - fl);
    const int lo = (ORDER & 1) ? 0 : 1;
    const int span = (ORDER & 1) ? in_len - 1 : in_len - 2;
    return (unsigned)(ci - lo) < (unsigned)max(span, 0);
}
// out-of-range axis: boundary map of the full coordinate c (deform.c:47-128, 782); returns true when
// the voxel maps to the constant
template <int ORDER, typename T>
__device__ __forceinline__ bool coord_axis_mapped(double c, int in_len, int mode, double period,
                                                  double inv_period, int& ci, T& frac)
{
    const double cc = map_coordinate_fast(c, in_len, mode, period, inv_period);
    const double fl = floor((ORDER & 1) ? cc : cc + 0.5);
    ci = (int)fl;
    frac = (T)(cc - fl);
    return !(cc > -1.0);
}

struct TileGeom {
    int in_len[3];        // I_k (the tile kernels require extents < 2^30)
    int out_len[3];
    int tiles[3];         // number of tiles per axis
    int strips_x;         // strips per tile row
    int strip_tiles;      // tiles per strip (8, or 4 / 2 for small outputs)
    int nstrips;
    int ncpx;             // control points along x = stride of one component row of Q
    // Wide control grids (more columns than the 64 Q rows of a strip can hold in LDS next to a box): the tables
    // kernel writes Q per x-STRIP -- only the q_win columns a strip of q_strip_vox voxels touches, starting at
    // the strip's lowest column -- as Q[o_z][o_y][strip][q_win][4], and the x-table's column indices relative to
    // that start.  The level-1 kernels of deform_hot.hip then see a grid of q_win columns; nobody else reads the
    // tables of such a call (launch_tile runs it in self-serve form).  q_win == 0: the plain layout.
    int q_win, q_strip_vox, q_strips;
    int box_cap;          // elements per LDS copy
    int off_ov;           // LDS byte offset of the overlay region (box | D, P)
    int in_stride[3];     // element strides (the tile kernels require < 2^31 elements per volume)
    int out_stride[3];
    int mode;             // boundary mode of this input
    int has_affine;
    int off[3];           // crop offsets
    double period[3];     // boundary-map period of each axis for `mode`, and its reciprocal
    double inv_period[3];
    double affine[12];    // inverse map, 3 x 4
    int* spill;           // [0] = count, [1..] = tile ids that did not fit in LDS
    int* spill_next;      // the second level's spill list (its count is reset by the tables kernel)
    // level-1 spill feedback (ed_workspace.h, SpillHint): hint[0] = tiles of this call whose box does not fit
    // the standard level-1 box, hint[1] = this call's sequence number; the tables kernel of the next call
    // on the stream reports the pair to hint_host (pinned host memory) before it resets them
    int* hint;
    unsigned long long* hint_host;
    unsigned hint_seq;
    // EDHIP_FLAG_ZERO_GRADIENT: the tables kernel's blocks beyond out_len[0] clear this block (16-byte
    // aligned start, any length)
    char* zero_ptr;
    long long zero_bytes;
    // forward -> gradient hand-over of the coordinate records (HotGeom::rec): the tables kernel of a
    // KEEP forward call copies the control-grid values it reads into keep_stash (keep_mode 1); the tables
    // kernel of a USE gradient call compares what it reads with the copy and writes keep_flags[sample]
    // = 1 when every value is bit-equal, 0 otherwise (keep_mode 2).  nbatch * 3 * prod(ncp) doubles.
    double* keep_stash;
    int* keep_flags;
    int keep_mode;
    int* label_list;      // label kernel: [0] = count, [1..cap] = linear ids of near-tie voxels (or nullptr)
    int label_cap;
    const int* worklist;  // second-level pass: [0] = count, [1..] = tile ids to process (else nullptr)
    // K1's sampled tile boxes (deform_k1.hip): block (0, sample) of the tables kernel leaves the margin they are
    // widened by, per component, in slack[sample * 4 + h] = clamp(slack_scale * max |D_f[h]|, 0.02, 0.75) voxels
    double* slack;
    double slack_scale;
    const double* q_global;   // [O_z][O_y][ncpx][4]: displacement contracted over z and y (component padded to 4)
    const AxTab* xt_global;   // [O_x]: cubic weights / control indices along x (x 4: Q element offsets)
    int dbg;              // ablation switches for profiling (EDHIP_TILE_DBG), 0 in production
    // batches of independent volumes with one control grid each (edhip_deform_batch): sample b of
    // nbatch lives at in + b * in_bstride, out + b * out_bstride (elements), its control grid at
    // disp + b * disp_bstride (bytes), its Q table at q_global + b * q_bstride (doubles); XT is shared.
    // Strip / tile ids carry the sample: id = sample * (strips | tiles per sample) + local id.
    int nbatch;
    int ntiles;           // tiles per sample
    long long in_bstride, out_bstride, q_bstride, disp_bstride;
};

struct StripPos {
    int tz, ty, tx0, ntile;   // tile coordinates of the strip and number of tiles in it
    int sample;               // which volume of the batch
};

__device__ __forceinline__ bool strip_position(const TileGeom& tg, int work, StripPos& sp)
{
    if (tg.worklist) {
        // second-level pass: one 8^3 tile per work item, taken from the first level's spill list
        if (work >= tg.worklist[0])
            return false;
        int t = tg.worklist[1 + work];
        sp.sample = t / tg.ntiles;
        t -= sp.sample * tg.ntiles;
        sp.tx0 = t % tg.tiles[2];
        t /= tg.tiles[2];
        sp.ty = t % tg.tiles[1];
        sp.tz = t / tg.tiles[1];
        sp.ntile = 1;
        return true;
    }
    if (work != (int)blockIdx.x)
        return false;        // first level: one strip per block
    // strips are dealt to the XCDs in contiguous chunks (block b runs on XCD b % 8): neighbouring
    // strips, whose source boxes overlap, share an L2
    const int b = blockIdx.x;
    const int total = tg.nstrips * tg.nbatch;
    const int per = (total + 7) >> 3;
    int s = (b & 7) * per + (b >> 3);
    if (s >= total)
        return false;
    sp.sample = s / tg.nstrips;
    s -= sp.sample * tg.nstrips;
    const int sx = s % tg.strips_x;
    s /= tg.strips_x;
    sp.ty = s % tg.tiles[1];
    sp.tz = s / tg.tiles[1];
    sp.tx0 = sx * tg.strip_tiles;
    sp.ntile = min(tg.strip_tiles, tg.tiles[2] - sp.tx0);
    return true;
}

// ---- the float32 benchmark kernels (deform_hot.hip) -------------------------------------------------
// Argument block of the hot kernels: 3 deformed axes, float32, unit stride along x on both sides.
// `vol` is the array with the INPUT's deformed extents (forward: the source volume, read; gradient:
// dX, accumulated into), `img` the one with the OUTPUT's extents (forward: written; gradient: dY).
// spill feedback of K2: a tile beyond the standard box counts 1, one beyond the LARGE box counts kHintHuge (the launcher
// reads "a tenth of the tiles beyond the large box" off the same counter: ed_workspace.h)
constexpr int kHintHuge = 64;
struct HotGeom {
    const float* vol_r;       // forward: source volume
    float* vol_w;             // gradient: dX
    const float* img_r;       // gradient: dY
    float* img_w;             // forward: destination
    const double* q;          // per-call tables (see tile_tables_kernel)
    const AxTab* xt;
    const double* slack;      // TileGeom::slack
    int* spill;               // tiles the LDS box cannot hold -> the general kernels
    int* hint;                // [0]: count of the tiles whose box exceeds small_cap elements (or nullptr)
    int small_cap;            // box_cap of the standard configuration (== box_cap unless the large boxes are in use)
    int large_cap;            // K2: box_cap of the large configuration -- a tile beyond it weighs kHintHuge in the count
    long long vol_bstride, img_bstride, q_bstride;      // elements between consecutive samples
    int in_len[3], out_len[3], off[3];
    int vol_sz, vol_sy;       // element strides of vol along z, y (x: 1)
    int img_sz, img_sy;
    int tiles[3];
    int strips_x, strip_tiles, nstrips, total_strips, ntiles;
    int ncpx, box_cap, off_box, mode, has_affine;
    int lds_grp;          // bytes of LDS per wave group (K2, two groups per workgroup)
    // tile bounding boxes handed from the forward to the gradient kernel: [tile][8] ints (lo x3,
    // hi x3, -, -), tile = sample * ntiles + (tz * tiles_y + ty) * tiles_x + tx.  K1 writes them when
    // the pointer is set; K2 reads them when `use_boxes` is set (see EDHIP_FLAG_USE_BOXES).
    int* boxes;
    int use_boxes;
    // Per-voxel coordinate records handed from the forward to the gradient call, next to the boxes
    // (deform_hot.hip, "records"): rec[sample * rec_bstride + (oz * O_y + oy) * O_x + ox] =
    // {frac_z, frac_y, frac_x, bits of the packed window start relative to the voxel's 8^3 tile box}.
    // K1 writes them when `rec` is set; rec_only: coordinates, boxes and records only (no staging, no
    // gather, no output) -- the first half of a gradient call that has no forward call to lean on.
    // rec_valid[sample] != 0 (device memory, written by the tables kernel of a gradient call): the
    // records in the buffer were made from these very displacement values -- a rec_only launch
    // leaves that sample alone.
    // self_serve: a geometry whose recent calls left (almost) no tile to the spill list (spill feedback,
    // ed_workspace.h) is launched without the level-2 / level-3 kernels behind it -- two launches that cost the
    // benchmark step 36 us for 2 tiles of 32768 -- and level 1 serves a tile whose box does not fit itself,
    // straight from / to global memory
    int self_serve;
    int io16;                     // IOView::out16: the output side (img_r / img_w) holds 16-bit floats
    int q_strips;                 // TileGeom::q_strips (1: the plain Q layout; ncpx then is the window width)
    float4* rec;
    long long rec_bstride;
    int rec_only;
    const int* rec_valid;
    unsigned long long* dbgbuf;   // EDHIP_EXPERIMENTS builds: per-workgroup timestamps (else unused)
    // integer fast path (wave_int_fwd_kernel): near-tie voxels for the exact re-evaluation,
    // [0] = count, [1 .. tie_cap] = linear output voxel ids; the constant of 'constant' mode in fp64
    int* tie_list;
    int tie_cap;
    double cvald;
    int dbg;                  // experiment switches (EDHIP_TILE_DBG), 0 in production
    float cval;
    int nstep;
    long long nsteps;
    long long step_len[8], vol_step[8], img_step[8];    // element strides of the step axes
    double period[3], inv_period[3], affine[12];
};

// coordinate records (HotGeom::rec): .w holds (start_z - box_z) | (start_y - box_y) << 8 | (start_x - box_x) << 16,
// each clamped to 255 (a tile with such a box is far beyond any LDS budget and goes to the general
// kernels, which do not read records), or kRecDead for a voxel that maps to the constant (deform.c:928)
constexpr unsigned kRecDead = 0x80000000u;

// level-1 launch of the hot kernels; hipErrorNotSupported when (order, ...) has no instantiation
hipError_t launch_hot_level1(const HotGeom& hg, int order, bool gradient, unsigned nblk, size_t lds,
                             hipStream_t stream);
// large: the configuration with one workgroup per CU fewer and larger boxes (chosen from the spill feedback)
// level: 0 standard, 1 large, 2 huge (gradient only: 64 KiB of cells, two workgroups per CU)
size_t hot_lds_bytes(bool gradient, int ncpx, int* box_cap, int* off_box, int level = 0);
// the records route of a gradient call: K1 in records-only form (grid / LDS of the forward launch), then the
// gradient kernel that reads records and boxes (orders 1-3)
hipError_t launch_hot_records(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream);
size_t hot_grad2_lds_bytes(int* box_cap, bool large = false);
hipError_t launch_hot_grad2(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream);

// K1 of round 5 (deform_k1.hip): float32 forward, orders 1-3, sampled tile boxes; same argument block, strips of at
// most 4 tiles; hipErrorNotSupported (nothing launched) otherwise
size_t k1_lds_bytes(int ncpx, int* box_cap, int* off_box, bool large = false);
hipError_t launch_k1_level1(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream);

// ---- K1 of round 6 (deform_k1z.hip): strips along z, displacement contracted over y and x ------------------------
// Second argument block of k1z_fwd_kernel / k1z_geo_kernel, next to HotGeom.
struct ZGeom {
    const double* r;          // [sample][O_y][O_x][ncp_z][4]: displacement contracted over y and x (component padded to 4)
    const AxTab* zt;          // [O_z]: cubic weights / control-plane byte offsets (x 32) along z
    int* recs;                // [sample * ntiles + tile][8]: tile records, written by the geometry kernel
    int* recs_half;           // [tile][2][8]: records of the z halves of the tiles whose whole box does not fit LDS
    int* missed;              // [sample * nstrips + strip]: a tile kernel saw a window outside its sampled box
    long long* steps;         // [nsteps][2]: element offsets (volume, image) of every index of the step axes
    int* sinfo;               // [sample * nstrips + strip]: 3 bits per tile of the strip (class, beyond the standard box)
    // work lists of strips: G = general tiles (geometry kernel), F = tiles that do not fit (geometry kernel) or a window
    // outside its sampled box (tile kernels).  ctl[parity] / ctl[2 + parity] = their counts for this call; the geometry
    // kernel clears the other parity's for the next call on the stream (the workspace head is cleared when allocated)
    int* list_g;
    int* list_f;
    int* ctl;
    int parity;
    void* zgen;               // 512 bytes: the uniform values of the general paths (ZGen), written by the geometry kernel
    int tables_only;          // the gradient route (deform_k2z.hip): R, the z table and ZGen, no records
    char* zero_ptr;           // EDHIP_FLAG_ZERO_GRADIENT: spare workgroups of the geometry launch clear this block
    long long zero_bytes;
    long long r_bstride;      // doubles between consecutive samples of r
    long long disp_bstride;   // bytes between the control grids of consecutive samples
    int ncpz;
    int order;
    int strip_tiles;          // tiles per strip (along z)
    int nstrips;              // strips per sample
    int total_strips;
    double slack_scale;       // margin of the sampled boxes = clamp(slack_scale * max |D_f[h]|, 0.02, 0.75)
    int* hint;                // spill feedback (TileGeom::hint): reset / reported by the geometry kernel, counted by K1
    unsigned long long* hint_host;
    unsigned hint_seq;
};
size_t k1z_lds_bytes(int* box_cap, bool large = false);
bool k1z_supported(const GridGeom& g);
size_t k1z_r_bytes(const GridGeom& g);
hipError_t launch_k1z_geo(const GridGeom& g, const HotGeom& hg, const ZGeom& zg, const GridPrefilter& gp, int nbatch,
                          hipStream_t stream);
hipError_t launch_k1z(const HotGeom& hg, const ZGeom& zg, int order, size_t lds, hipStream_t stream, ed::SideLane* side);
// K2 of round 6 (deform_k2z.hip): the gradient on the z-walk tables (geometry kernel in its tables-only form)
size_t k2z_lds_bytes(int* box_cap);
hipError_t launch_k2z(const HotGeom& hg, const ZGeom& zg, int order, size_t lds, hipStream_t stream);

// one-wavefront-per-tile kernels (deform_wave.hip): same argument block; `strip_tiles`, `strips_x`,
// `nstrips`, `total_strips` describe the strips of the 64-thread workgroups, `box_cap` the floats /
// cells of LDS one wave owns
hipError_t launch_wave_level1(const HotGeom& hg, int order, bool gradient, unsigned nblk, size_t lds, int occ,
                              hipStream_t stream);
size_t wave_lds_bytes(bool gradient, int occ, int* box_cap);
// integer volumes (8- / 16-bit), orders 1-5, forward: fast coordinates, fp64 taps, near-tie voxels listed
// for the exact kernel; `dtype` is the edhip_dtype of input and output
hipError_t launch_wave_int(const HotGeom& hg, int order, int dtype, const void* vol, void* img, unsigned nblk,
                           size_t lds, hipStream_t stream);

}  // namespace tile
}  // namespace ed
