// deform_tile.hip -- K1 (forward gather) and K2 (gradient scatter-add), LDS-tiled: the hot kernels
// of the benchmark workload (3 deformed axes, float32 / float64 volumes, spline orders 1-5), plus
// the order-0 direct kernel and the label kernel (order 0, every dtype, bit-equal to the exact path).
//
// Per-voxel pipeline of DeformGrid's hot loop (deform.c:649-1001), organised per OUTPUT TILE:
//
//   tables   (tile_tables_kernel, once per call) the displacement spline is a tensor product
//            (deform.c:639-758), so it is contracted over z and y once per call, in fp64:
//            Q[o_z][o_y][k_x][c] = sum w_z w_y D_f, plus the x-table XT[o_x] (cubic weights and
//            mirror-mapped control indices -- the reference's `dsplvals`).  A voxel is left with 4
//            x-taps per component: 12 fp64 FMAs instead of the reference's 192 multiply-adds.
//   strip    a 256-thread workgroup (4 waves) owns a strip of up to 8 tiles along x (4 / 2 for small
//            outputs); a tile is 8 x 8 x 8 output voxels (cubic, so the source bounding box stays
//            small under shear: SURVEY.md section 7 -- long-x tiles overfetch 6x, cubes 3.5x); a lane
//            owns one (y, x) column of the tile and two of its z-slices (K2: 8 x 8 x 16 tiles, four
//            voxels per lane).  The strip prologue copies its 64 Q rows and 64 XT entries into LDS.
//   phase A  per tile: displacement, affine, + offset, boundary map, floor -- all fp64 -- then the
//            fractional offsets are handed to the data's width for the basis weights.
//   phase B  bounding box of all tap windows of the tile in UNMAPPED tap-index space (wave
//            min/max reduce, one LDS atomic per wave; three result slots in rotation).
//   K1 C/D   the source box is staged from HBM/L2 into LDS once, rows coalesced along the fastest
//            axis; every box index goes through the mirror map here, which is what the reference
//            does with the taps of a window that sticks out (deform.c:791-813) -- the gather needs
//            no edge handling.  float32: a second copy shifted by one element makes every x-run of
//            taps aligned ds_read_b64 pairs (even orders pad a zero-weight tap).  Then the
//            (order+1)^3 tap gather from LDS, accumulated separably (x, y, z) in the data's width;
//            streaming output stores.
//   K2 C/D   the box is an accumulator: taps are scattered with INTEGER LDS atomics (ds_add_u32
//            sustains ~5 cycles per wave instruction on MI355X, ds_add_f32 ~195 --
//            profiles/r01_ubench_lds.txt; float64 volumes use int64 cells) in a per-tile
//            fixed-point scale derived from the tile's sum of |dY|, which cannot overflow and
//            resolves each contribution far below the float rounding of the reference's own `+=`
//            (deform.c:309-312).  The box is then flushed with one float atomic per touched source
//            element (mirror-mapped; dX must be zero on entry): ~2.3 global atomics per voxel
//            instead of 64.
//   spill    a tile whose box exceeds the first-level LDS budget (strong folding, 'wrap' seams) is
//            appended to a worklist: level 2 retries it alone with a 48 KiB box, level 3 (the
//            direct kernel) finishes what is left straight from global memory with per-tap mirror
//            mapping; the hot kernels carry no fallback code.
//
// HBM traffic (256^3 float32, profiles/hbm_traffic.json): 61 MB fetched -- the source volume once;
// neighbouring tiles overlap, but strips are dealt to the 8 XCDs in contiguous chunks so that the
// overlap stays inside one L2 -- and 85 MB written; algorithmic bytes are 4 + 4 per voxel.
#include <atomic>
#include <type_traits>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ed_device.h"
#include "ed_exact_coord.h"
#include "ed_params.h"
#include "ed_gridfilter.h"
#include "ed_workspace.h"
#include "ed_tile.h"

#ifdef EDHIP_EXPERIMENTS
// profiling build: control columns the per-strip Q window of a wide grid (TileGeom::q_win) did not hold -- the
// tables kernel clamps them silently; tests/fuzz/fuzz_round4.py asserts the count stays 0 (edhip_debug_wide_clamped)
__device__ unsigned g_wide_clamped;
extern "C" unsigned edhip_debug_wide_clamped(void)
{
    unsigned v = 0xffffffffu, z = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wide_clamped), sizeof(v)) != hipSuccess)
        return 0xffffffffu;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wide_clamped), &z, sizeof(z));
    return v;
}
#endif

namespace ed {

namespace {

using namespace tile;

// Per-call tables (one small launch before the tile kernel).  The displacement spline is
// separable and its weights depend only on the output index along each axis (deform.c:639-647):
//   XT[ox]            = cubic weights + mirror-mapped control indices along x   (the reference's
//                       `dsplvals` rows, for one axis)
//   Q[oz][oy][j2][h]  = sum_{l0,l1} wz[oz][l0] wy[oy][l1] D[h, iz[l0], iy[l1], j2]   (h padded to 4:
//                       one control column is 32 bytes, XT's indices are pre-multiplied by 4)
// so that a voxel is left with 4 x-taps per component (12 fp64 FMAs instead of the reference's 192
// multiply-adds, deform.c:693-758).  One block per output z: P = D contracted over z in LDS, then
// every output y of that slice.
// gp.total > 0: the control grid at gp.in is still RAW -- every workgroup filters its own LDS copy first
// (ed_gridfilter.h: the bits of grid_prefilter_kernel, ~1 us for a 5^3 grid) and workgroup 0 writes the filtered
// grid to gp.out (= g.disp) for the kernels behind this one; saves the launch in front of this one.
__global__ __launch_bounds__(kBlock) void tile_tables_kernel(const GridGeom g, const TileGeom tg, const GridPrefilter gp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sP = reinterpret_cast<double*>(smem);              // [3][nyx]
    __shared__ AxTab tz_;
    const int tid = threadIdx.x;
    const int oz = blockIdx.x;
    const int sample = blockIdx.y;        // one control grid (and one Q table) per volume of a batch
    if (oz >= tg.out_len[0]) {
        // spare workgroups: clear the gradient accumulators (EDHIP_FLAG_ZERO_GRADIENT) while the others
        // compute the tables -- 64 MB in ~15 us next to an 8 us kernel instead of a 21 us launch in front of it
        const long long nfill = (long long)(gridDim.x - tg.out_len[0]) * kBlock;
        const long long me = (long long)(oz - tg.out_len[0]) * kBlock + tid;
        const long long n16 = tg.zero_bytes >> 4;
        int4* p16 = reinterpret_cast<int4*>(tg.zero_ptr);
        for (long long i = me; i < n16; i += nfill)
            p16[i] = make_int4(0, 0, 0, 0);
        if (me < (tg.zero_bytes & 15))
            tg.zero_ptr[(n16 << 4) + me] = 0;
        return;
    }
    const char* disp = g.disp + (int64_t)sample * tg.disp_bstride;
    const int ncpy = (int)g.ncp[1], ncpx = (int)g.ncp[2];
    const int nyx = ncpy * ncpx;
    const bool own = gp.total > 0;
    double* sG = sP + 3 * nyx;                                 // [3][ncp_z][nyx], own only
    if (own) {
        grid_prefilter_in_lds<kBlock>(gp, sG, tid);
        if (oz == 0)
            for (int e = tid; e < gp.total; e += kBlock)
                store_cast(gp.out + (int64_t)e * gp.elem_size, gp.dtype, sG[e]);
    }
    // control coefficient D_f[h][j0][j1][j2]
    auto dval = [&](int h, int j0, int j1, int j2) -> double {
        if (own)
            return sG[((h * (int)g.ncp[0] + j0) * ncpy + j1) * ncpx + j2];
        return load_as_double(disp + g.disp_stride[0] * h + g.disp_stride[1] * j0 + g.disp_stride[2] * j1 +
                                  g.disp_stride[3] * j2, g.disp_dtype);
    };
    if (oz == 0 && sample == 0 && tid == 0) {      // reset both spill counters (saves two memset launches per call)
        if (tg.hint) {
            // the previous call on this stream: (sequence number, tiles beyond the standard box) -> the host
            if (tg.hint_host)
                __hip_atomic_store(tg.hint_host, ((unsigned long long)(unsigned)tg.hint[1] << 32) | (unsigned)tg.hint[0],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            tg.hint[0] = 0;
            tg.hint[1] = (int)tg.hint_seq;
        }
        tg.spill[0] = 0;
        tg.spill_next[0] = 0;
        if (tg.label_list)
            tg.label_list[0] = 0;
    }
    if (oz == 0 && tg.slack) {
        // margin of K1's sampled tile boxes (deform_k1.hip): a fraction of the rigorous bound of multilinear
        // interpolation between samples <= 3 voxels apart, 9/8 * sum_k r_k^2 max |second difference| with
        // |second difference| <= 4 max |D_f|; slack_scale holds everything but the maximum
        __shared__ double smax[3][kBlock / 64];
        const int ntot = (int)g.ncp[0] * nyx;
        for (int h = 0; h < 3; ++h) {
            double m = 0.0;
            for (int e = tid; e < ntot; e += kBlock) {
                const int j0 = e / nyx, j = e - j0 * nyx;
                const int j1 = j / ncpx, j2 = j - j1 * ncpx;
                m = fmax(m, fabs(dval(h, j0, j1, j2)));
            }
            for (int sh = 32; sh >= 1; sh >>= 1)
                m = fmax(m, __shfl_xor(m, sh));
            if ((tid & 63) == 0)
                smax[h][tid >> 6] = m;
        }
        __syncthreads();
        if (tid < 3) {
            double m = 0.0;
            for (int w = 0; w < kBlock / 64; ++w)
                m = fmax(m, smax[tid][w]);
            const double sl = tg.slack_scale * m;
            tg.slack[sample * 4 + tid] = sl >= 0.02 ? (sl <= 0.75 ? sl : 0.75) : 0.02;     // (NaN -> 0.02)
        }
    }
    auto entry = [&](int a, int oi, AxTab& t) {
        const double cp = control_coordinate(g.ncp[a], (int64_t)oi + g.off[a], g.in_len[a]);
        const int64_t start = window_start(cp, 3);
        const bool edge = start < 0 || start + 3 >= g.ncp[a];
        double w[4];
        spline_weights(cp, 3, w);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            t.w[l] = w[l];
            // (32-bit mirror map: a 64-bit integer division costs ~1 us here, and with few control
            // points most windows are edge windows)
            t.idx[l] = edge ? mirror_i32((int)start + l, (int)g.ncp[a]) : (int)start + l;
        }
    };
    if (tid == 0)
        entry(0, oz, tz_);
    if (oz == 0 && tg.keep_mode) {
        // the control-grid values this call works from: kept for (1) / compared with (2) the gradient
        // call that wants to use the forward call's coordinate records
        const int nz = (int)g.ncp[0];
        const int ntot = 3 * nz * nyx;
        double* stash = tg.keep_stash + (int64_t)sample * ntot;
        int same = 1;
        for (int e = tid; e < ntot; e += kBlock) {
            const int h = e / (nz * nyx), r = e - h * (nz * nyx);
            const int j0 = r / nyx, j = r - j0 * nyx;
            const int j1 = j / ncpx, j2 = j - j1 * ncpx;
            const double val = dval(h, j0, j1, j2);
            if (tg.keep_mode == 1) {
                stash[e] = val;
            } else if (__double_as_longlong(stash[e]) != __double_as_longlong(val)) {
                same = 0;
                stash[e] = val;       // the records-only launch behind this kernel remakes the records from `val`
            }
        }
        if (tg.keep_mode == 2) {
            same = __syncthreads_and(same);
            if (tid == 0)
                tg.keep_flags[sample] = same;
        }
    }
    // wide control grids (TileGeom::q_win): the lowest control column each x-strip touches
    __shared__ int c0s[256];
    if (tg.q_win) {
        for (int sidx = tid; sidx < tg.q_strips; sidx += kBlock)
            c0s[sidx] = 0x7fffffff;
        __syncthreads();
        for (int ox = tid; ox < tg.out_len[2]; ox += kBlock) {
            AxTab t;
            entry(2, ox, t);
            const int lo = min(min(t.idx[0], t.idx[1]), min(t.idx[2], t.idx[3]));
            atomicMin(&c0s[ox / tg.q_strip_vox], lo);
        }
        __syncthreads();
    }
    if (oz == 0 && sample == 0) {
        AxTab* xt = const_cast<AxTab*>(tg.xt_global);
        for (int ox = tid; ox < tg.out_len[2]; ox += kBlock) {
            AxTab t;
            entry(2, ox, t);
            const int c0 = tg.q_win ? c0s[ox / tg.q_strip_vox] : 0;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                // element offset of control column idx in a Q row [ncpx][4] ([q_win][4], from the strip's lowest
                // column: the host sized q_win for every strip; the clamp only keeps a wrong size inside the row)
                const int rel = t.idx[l] - c0;
#ifdef EDHIP_EXPERIMENTS
                if (tg.q_win && (rel < 0 || rel >= tg.q_win))
                    atomicAdd(&g_wide_clamped, 1u);      // (the host's window formula was too small: must never count)
#endif
                t.idx[l] = (tg.q_win ? min(rel, tg.q_win - 1) : rel) * 4;
            }
            xt[ox] = t;
        }
    }
    __syncthreads();
    for (int e = tid; e < 3 * nyx; e += kBlock) {
        const int h = e / nyx, j = e - h * nyx;
        const int j1 = j / ncpx, j2 = j - j1 * ncpx;
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l)
            acc += tz_.w[l] * dval(h, tz_.idx[l], j1, j2);
        sP[e] = acc;
    }
    __syncthreads();
    double* q = const_cast<double*>(tg.q_global) + (int64_t)sample * tg.q_bstride;
    if (tg.q_win) {
        // per strip: q_win columns from the strip's lowest one (columns beyond the grid: zero, never read)
        const int W = tg.q_win;
        for (int oy = tid; oy < tg.out_len[1]; oy += kBlock) {
            AxTab ty;
            entry(1, oy, ty);
            double* row = q + ((int64_t)oz * tg.out_len[1] + oy) * tg.q_strips * (4 * W);
            for (int sidx = 0; sidx < tg.q_strips; ++sidx) {
                const int c0 = c0s[sidx];
                for (int k = 0; k < W; ++k) {
                    const int j2 = c0 + k;
                    double acc[3] = {0.0, 0.0, 0.0};
                    if (j2 < ncpx) {
#pragma unroll
                        for (int h = 0; h < 3; ++h) {
#pragma unroll
                            for (int l = 0; l < 4; ++l)
                                acc[h] += ty.w[l] * sP[h * nyx + ty.idx[l] * ncpx + j2];
                        }
                    }
                    double* dst = row + ((int64_t)sidx * W + k) * 4;
                    dst[0] = acc[0];
                    dst[1] = acc[1];
                    dst[2] = acc[2];
                    dst[3] = 0.0;
                }
            }
        }
        return;
    }
    for (int oy = tid; oy < tg.out_len[1]; oy += kBlock) {
        AxTab ty;
        entry(1, oy, ty);
        double* row = q + ((int64_t)oz * tg.out_len[1] + oy) * 4 * ncpx;
        for (int j2 = 0; j2 < ncpx; ++j2) {
            for (int h = 0; h < 3; ++h) {
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    acc += ty.w[l] * sP[h * nyx + ty.idx[l] * ncpx + j2];
                row[j2 * 4 + h] = acc;
            }
            row[j2 * 4 + 3] = 0.0;
        }
    }
}

// Strip prologue: the strip's 64 x-table entries and its 64 rows of Q come from the per-call
// tables into LDS; hot uniform parameters are parked in LDS.  Ends with a barrier.
__device__ __forceinline__ void strip_prologue(const GridGeom& g, const IOView& vstep,
                                               const TileGeom& tg, const StripPos& sp, char* smem)
{
    int* sred = reinterpret_cast<int*>(smem + kOffRed);
    double* sQ = reinterpret_cast<double*>(smem + kOffQ);
    const int tid = threadIdx.x;
    (void)g;

    {   // x table: 64 entries x 48 bytes = 768 dwords
        const int* src = reinterpret_cast<const int*>(tg.xt_global + sp.tx0 * kT);
        int* dst = reinterpret_cast<int*>(smem + kOffTabX);
        const int avail = (tg.out_len[2] - sp.tx0 * kT) * 12;      // dwords that exist
        for (int e = tid; e < kStrip * kT * 12; e += kBlock)
            dst[e] = e < avail ? src[e] : 0;
    }
    {   // Q rows: (zi, yy) -> global row (oz, oy); 4 threads per row
        const int rowlen = 4 * tg.ncpx;
        const int r = tid >> 2;
        const int oz = min(sp.tz * kT + (r >> 3), tg.out_len[0] - 1);
        const int oy = min(sp.ty * kT + (r & 7), tg.out_len[1] - 1);
        // (wide control grids: Q is laid out per x-strip, tg.ncpx = q_win columns each -- TileGeom::q_win)
        const int64_t qrow_id = tg.q_win ? ((int64_t)oz * tg.out_len[1] + oy) * tg.q_strips + (sp.tx0 * kT) / tg.q_strip_vox
                                         : (int64_t)oz * tg.out_len[1] + oy;
        const double* src = tg.q_global + (int64_t)sp.sample * tg.q_bstride + qrow_id * rowlen;
        double* dst = sQ + r * rowlen;
        for (int k = tid & 3; k < rowlen; k += 4)
            dst[k] = src[k];
    }
    if (tid < 24) {
        const int k = tid & 7;
        sred[tid] = k < 3 ? 0x7fffffff : (int)0x80000000;
    }
    if (tid >= 128 && tid < 128 + 12) {
        HotParams* hp = reinterpret_cast<HotParams*>(smem + kOffHot);
        const int k = tid - 128;
        hp->affine[k] = tg.affine[k];
        if (k < 3) {
            hp->offd[k] = (double)tg.off[k];
            hp->last[k] = (double)(tg.in_len[k] - 1);
            hp->period[k] = tg.period[k];
            hp->inv_period[k] = tg.inv_period[k];
        }
        if (k < 8) {
            hp->step_len[k] = vstep.step_len[k];
            hp->in_step_stride[k] = vstep.in_step_stride[k];
            hp->out_step_stride[k] = vstep.out_step_stride[k];
        }
        if (k == 0)
            hp->nstep = vstep.nstep;
    }
    __syncthreads();
}

// Phase A for one voxel: displacement from Q, affine, boundary map, window start, fraction.
template <typename T, int ORDER>
__device__ __forceinline__ bool voxel_coords(const TileGeom& tg, const HotParams* hp, const double* sQ,
                                             const AxTab& tx_, int zi, int yy, const int* o,
                                             int* start, T* frac)
{
    double c[3];          // affine: the full coordinate; otherwise the displacement alone
    int bi[3], ci[3];
    bool inr[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const double* qrow = sQ + (zi * kT + yy) * 4 * tg.ncpx + h;
        double d = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l)
            d = fma(tx_.w[l], qrow[tx_.idx[l]], d);
        if (tg.has_affine) {
            double b = hp->affine[h * 4 + 3];
#pragma unroll
            for (int l = 0; l < 3; ++l)
                b = fma(hp->affine[h * 4 + l], (double)o[l], b);
            c[h] = b + hp->offd[h] + d;
            bi[h] = 0;
        } else {
            c[h] = d;
            bi[h] = o[h] + tg.off[h];               // the crop offset folds into the integer
        }
        inr[h] = coord_axis_fast<ORDER, T>(c[h], bi[h], tg.in_len[h], ci[h], frac[h]);
    }
    bool cst = false;
    if (!(inr[0] && inr[1] && inr[2])) {
        // one divergent region: the axes along which the source point left the array
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            if (!inr[h])
                cst = coord_axis_mapped<ORDER, T>((double)bi[h] + c[h], (int)hp->last[h] + 1, tg.mode,
                                                  hp->period[h], hp->inv_period[h], ci[h], frac[h]) || cst;
        }
    }
#pragma unroll
    for (int h = 0; h < 3; ++h)
        start[h] = cst ? 0 : ci[h] - ORDER / 2;
    return cst;
}

__device__ __forceinline__ void step_offsets(const HotParams* hp, int64_t ss, int64_t& in_off,
                                             int64_t& out_off)
{
    in_off = 0;
    out_off = 0;
    int64_t r = ss;
    const int nstep = hp->nstep;
    for (int l = 0; l < nstep; ++l) {
        const int64_t len = hp->step_len[l];
        const int64_t q = r / len;
        const int64_t c = r - q * len;
        in_off += hp->in_step_stride[l] * c;
        out_off += hp->out_step_stride[l] * c;
        r = q;
    }
}

__device__ __forceinline__ void step_offsets(const IOView& v, int64_t ss, int64_t& in_off,
                                             int64_t& out_off)
{
    in_off = 0;
    out_off = 0;
    int64_t r = ss;
    for (int l = 0; l < v.nstep; ++l) {
        const int64_t q = r / v.step_len[l];
        const int64_t c = r - q * v.step_len[l];
        in_off += v.in_step_stride[l] * c;
        out_off += v.out_step_stride[l] * c;
        r = q;
    }
}

// One tap of the separable gather.  Every forward kernel of the tile path -- the 4-wave and the
// one-wave kernels of level 1 (deform_hot.hip, deform_wave.hip), the general kernels of levels 1 / 2 and
// the direct kernel of level 3 -- accumulates x first, then y, then z, each as a chain of fused
// multiply-adds starting from zero: a voxel gets the same bits whichever level serves its tile, so
// results do not depend on the box configuration (spill feedback), on the batch a sample travels in,
// or on a crop.  (Left to the compiler's contraction, `a += w * x` came out fused in some kernels
// and not in others: 2-9 % of the voxels differed by an ulp between levels.)
__device__ __forceinline__ float tap_fma(float w, float x, float a) { return __builtin_fmaf(w, x, a); }
__device__ __forceinline__ double tap_fma(double w, double x, double a) { return __builtin_fma(w, x, a); }

// ================================================================================================
// K1: forward
// ================================================================================================
// (Tried: a level-2 launch that finds only a handful of tiles on its list -- 2 of 32768 for the benchmark
// volume -- passes them on to the direct level instead of spending a ~14 us launch on them.  The direct
// kernel evaluates the displacement spline from the control grid for every voxel: a lone tile takes it
// longer than level 2 does; 256^3 forward call 250 -> 263 us.)
// ABL: compile-time ablation switches for profiling (0 in production): 1 skip staging loads,
// 2 skip gather, 4 skip coordinates, 16 prologue only, 32 skip staging entirely, 64 skip the output store
// (Tried: level 3 folded into the level-2 kernels -- the direct path called where a tile overflows
// the 48 KiB box too; one 4 us launch less per call, but the level-2 forward kernel loses registers
// to it: 159 -> 198 us at sigma = 10, where 6000 tiles take this path.  And the grid prefilter fused
// into the tables kernel (every block filters its own LDS copy of a small raw grid; one 8 us launch
// less): no measurable change of the 256^3 step in an A/B on one box, 0.7935 against 0.7961 ms.)
template <typename T, int ORDER, bool PAIR, int ABL = 0>
__global__ __launch_bounds__(kBlock, 4) void deform_tile3_fwd_kernel(const GridGeom g,
                                                                     const IOView v,
                                                                     const TileGeom tg)
{
    constexpr int NT = ORDER + 1;
    // PAIR reads the x-taps as aligned pairs: even orders (an odd number of taps) read one padding
    // element with weight zero, and the box is one element wider so that the read stays inside it
    constexpr int kPadX = (PAIR && (NT & 1)) ? 1 : 0;
    constexpr int NTX = NT + kPadX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int work = blockIdx.x;; work += gridDim.x) {
    StripPos sp;
    if (!strip_position(tg, work, sp))
        return;
    if (work != (int)blockIdx.x)
        __syncthreads();      // the previous work item is done with the LDS
    strip_prologue(g, v, tg, sp, smem);

    const AxTab* tabx = reinterpret_cast<const AxTab*>(smem + kOffTabX);
    int* sred = reinterpret_cast<int*>(smem + kOffRed);
    const double* sQ = reinterpret_cast<const double*>(smem + kOffQ);
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem + kOffHot);
    T* box0 = reinterpret_cast<T*>(smem + tg.off_ov);
    T* box1 = box0 + tg.box_cap;          // cap = 56 (mod 64): the two copies sit on disjoint LDS banks

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int yy = lane >> 3, xx = lane & 7;
    const T* __restrict__ in = reinterpret_cast<const T*>(v.in) + (int64_t)sp.sample * tg.in_bstride;
    T* out = reinterpret_cast<T*>(v.out) + (int64_t)sp.sample * tg.out_bstride;
    if (ABL & 16)
        return;

    for (int ti = 0; ti < sp.ntile; ++ti) {
        const int o0[3] = {sp.tz * kT, sp.ty * kT, (sp.tx0 + ti) * kT};
        int* red = sred + (ti % 3) * 8;

        // ---- phase A: coordinates of this thread's two voxels ----------------------------------
        int start[2][3];
        T frac[2][3];
        bool valid[2], constant[2];
        int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int zi = wave + 4 * i;
            const int o[3] = {o0[0] + zi, o0[1] + yy, o0[2] + xx};
            valid[i] = o[0] < tg.out_len[0] && o[1] < tg.out_len[1] && o[2] < tg.out_len[2];
            if (ABL & 4) {
                constant[i] = false;
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    start[i][h] = min(max(o[h] - 1, 0), tg.in_len[h] - 4);
                    frac[i][h] = (T)0.5;
                }
            } else
            constant[i] = voxel_coords<T, ORDER>(tg, hp, sQ, tabx[ti * kT + xx], zi, yy, o, start[i],
                                                 frac[i]);
            if (valid[i] && !constant[i]) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    lo[h] = min(lo[h], start[i][h]);
                    hi[h] = max(hi[h], start[i][h] + ORDER + (h == 2 ? kPadX : 0));
                }
            }
        }
        // ---- phase B: bounding box of the tile's tap windows -----------------------------------
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int l = wave_min(lo[h]);
            const int u = wave_max(hi[h]);
            if (lane == 0) {
                atomicMin(&red[h], l);
                atomicMax(&red[3 + h], u);
            }
        }
        __syncthreads();   // B1: box known; every gather of the previous tile is done
        const int b0[3] = {red[0], red[1], red[2]};
        const int ext[3] = {red[3] - red[0] + 1, red[4] - red[1] + 1, red[5] - red[2] + 1};
        const bool any = red[3] >= red[0];
        // Re-arm the buffer of tile ti + 2 (three buffers in rotation).  Not the next tile's: a tile
        // that is handed to the spill list, or has no live voxel, has no second barrier, so a fast
        // wave can already be reducing tile ti + 1 into its buffer while this store is pending.
        // Tile ti + 2's buffer is safe: nobody gets to reduce into it before passing B1 of tile
        // ti + 1, which this thread only reaches after the store; and its last readers (tile
        // ti - 1, right after that tile's B1) are all past this tile's B1.
        if (tid < 6)
            sred[((ti + 2) % 3) * 8 + tid] = tid < 3 ? 0x7fffffff : (int)0x80000000;
        // row pitch: PAIR (ds_read_b64): 16 * odd puts 4 consecutive rows on 4 disjoint bank
        // groups; b32 / f64 reads: 8 or 24 (mod 32)
        int pitch;
        if (PAIR)
            pitch = ext[2] <= 16 ? 16 : (ext[2] <= 48 ? 48 : 0);
        else
            pitch = ext[2] <= 8 ? 8 : (ext[2] <= 24 ? 24 : (ext[2] <= 56 ? 56 : 0));
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const bool fits = pitch > 0 && (int64_t)nrows * pitch <= tg.box_cap;
        if (any && !fits) {
            if (tid == 0) {    // hand the whole tile to the spill kernel
                const int slot = atomicAdd(&tg.spill[0], 1);
                tg.spill[1 + slot] = sp.sample * tg.ntiles +
                                     (sp.tz * tg.tiles[1] + sp.ty) * tg.tiles[2] + sp.tx0 + ti;
            }
            continue;
        }
        const bool x_inside = b0[2] >= 0 && b0[2] + ext[2] <= tg.in_len[2];
        // the whole padded row (and its one-element shift) lies inside the line: vector staging
        const bool wide = tg.in_stride[2] == 1 && b0[2] >= 0 && b0[2] + pitch + 1 <= tg.in_len[2] &&
                          !(ABL & 1);
        const float inv_by = 1.0f / (float)by;

        for (int64_t ss = 0; ss < v.nsteps; ++ss) {
            int64_t in_off, out_off;
            step_offsets(hp, ss, in_off, out_off);
            const T* src = in + in_off;

            if (any && !(ABL & 32)) {
                // ---- phase C: stage the source box (mirror-mapped) into LDS ----------------------
                if (ss > 0)
                    __syncthreads();     // previous step's gathers are done with the box
                if (sizeof(T) == 4 && wide) {
                    // interior tile, unit x-stride: 16-byte chunks.  A thread moves chunk q of row
                    // r (4 / 12 chunks per row, consecutive lanes -> consecutive chunks: global
                    // reads are 64-byte runs, LDS writes conflict-free b128); the shifted copy is a
                    // second, 4-byte-offset load of the same run.
                    const int cpr = pitch >> 2;              // 2, 4, 6, 10, 12 or 14 chunks per row
                    const int total = nrows * cpr;
                    const float inv_cpr = 1.0f / (float)cpr;
                    for (int idx = tid; idx < total; idx += kBlock) {
                        const int r = cpr == 4 ? idx >> 2 : (int)(((float)idx + 0.5f) * inv_cpr);
                        const int q = idx - r * cpr;
                        const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                        const int zs = mirror_i32(b0[0] + zr, tg.in_len[0]);
                        const int ys = mirror_i32(b0[1] + yr, tg.in_len[1]);
                        const float* rowp = reinterpret_cast<const float*>(src) +
                                            (zs * tg.in_stride[0] + ys * tg.in_stride[1] + b0[2] + 4 * q);
                        const F4u v0 = *reinterpret_cast<const F4u*>(rowp);
                        float* d0 = reinterpret_cast<float*>(box0) + r * pitch + 4 * q;
                        *reinterpret_cast<float4*>(d0) = make_float4(v0.x, v0.y, v0.z, v0.w);
                        if (PAIR) {
                            const F4u v1 = *reinterpret_cast<const F4u*>(rowp + 1);
                            float* d1 = reinterpret_cast<float*>(box1) + r * pitch + 4 * q;
                            *reinterpret_cast<float4*>(d1) = make_float4(v1.x, v1.y, v1.z, v1.w);
                        }
                    }
                } else {
                const int sub = tid & 7;
                for (int r = tid >> 3; r < nrows; r += kBlock / 8) {
                    const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                    const int zs = mirror_i32(b0[0] + zr, tg.in_len[0]);
                    const int ys = mirror_i32(b0[1] + yr, tg.in_len[1]);
                    const T* rowp = src + (zs * tg.in_stride[0] + ys * tg.in_stride[1]);
                    T* d0 = box0 + r * pitch;
                    T* d1 = box1 + r * pitch;
                    for (int xi = sub; xi < ext[2]; xi += 8) {
                        const int xs = x_inside ? b0[2] + xi : mirror_i32(b0[2] + xi, tg.in_len[2]);
                        const T val = (ABL & 1) ? (T)xs : rowp[xs * tg.in_stride[2]];
                        d0[xi] = val;
                        if (PAIR && xi > 0)
                            d1[xi - 1] = val;
                    }
                }
                }
                __syncthreads();         // B2
            }

            // ---- phase D: gather -------------------------------------------------------------------
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!valid[i])
                    continue;
                T val;
                if (constant[i]) {
                    val = (T)v.cval;
                } else if (ABL & 2) {
                    val = frac[i][0] + frac[i][1] + frac[i][2] + (T)start[i][0];
                } else {
                    T w0[NT], w1[NT], w2[NTX];
                    weights_from_frac<T, ORDER>(frac[i][0], w0);
                    weights_from_frac<T, ORDER>(frac[i][1], w1);
                    weights_from_frac<T, ORDER>(frac[i][2], w2);
                    if (kPadX)
                        w2[NT] = (T)0;
                    const int rz = start[i][0] - b0[0], ry = start[i][1] - b0[1],
                              rx = start[i][2] - b0[2];
                    const int rowbase = (rz * by + ry) * pitch;
                    T a0 = 0;
                    if (PAIR) {
                        // consecutive x-taps as aligned 8-byte reads from the copy whose shift
                        // matches the parity of rx
                        const T* bp = (rx & 1) ? box1 + rowbase + rx - 1 : box0 + rowbase + rx;
#pragma unroll
                        for (int l0 = 0; l0 < NT; ++l0) {
                            T a1 = 0;
#pragma unroll
                            for (int l1 = 0; l1 < NT; ++l1) {
                                const T* rp = bp + (l0 * by + l1) * pitch;
                                T a2 = 0;
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
                    } else {
                        const T* bp = box0 + rowbase + rx;
#pragma unroll
                        for (int l0 = 0; l0 < NT; ++l0) {
                            T a1 = 0;
#pragma unroll
                            for (int l1 = 0; l1 < NT; ++l1) {
                                const T* rp = bp + (l0 * by + l1) * pitch;
                                T a2 = 0;
#pragma unroll
                                for (int l2 = 0; l2 < NT; ++l2)
                                    a2 = tap_fma(w2[l2], rp[l2], a2);
                                a1 = tap_fma(w1[l1], a2, a1);
                            }
                            a0 = tap_fma(w0[l0], a1, a0);
                        }
                    }
                    val = a0;
                }
                const int oz = o0[0] + wave + 4 * i, oy = o0[1] + yy, ox = o0[2] + xx;
                T* optr = out + (out_off + (oz * tg.out_stride[0] + oy * tg.out_stride[1] + ox * tg.out_stride[2]));
                // streaming store: a tile writes 32-byte row segments, the rest of each 128-byte
                // line arrives tiles later; with a cached store the L2 evicts and refills the
                // half-written lines (measured: +49 MB fetched, +55 MB written per 256^3 launch)
                if (!(ABL & 64) || val == (T)-12345.678)
                    __builtin_nontemporal_store(val, optr);
            }
        }
    }
    }   // work items
}

// ================================================================================================
// K2: gradient (float32 / float64): integer LDS accumulation per tile, float atomics to flush
// ================================================================================================
// fixed-point accumulator of the data type: float32 -> int32, float64 -> int64 (ds_add_u64)
template <typename T>
struct GradFixed;
template <>
struct GradFixed<float> {
    typedef int acc_t;
    typedef unsigned int uacc_t;
    static constexpr double kRange = 2147483648.0 - 1024.0;
    __device__ static acc_t round(float x) { return __float2int_rn(x); }
    __device__ static bool finite(float x) { return (__float_as_int(x) & 0x7f800000) != 0x7f800000; }
};
template <>
struct GradFixed<double> {
    typedef long long acc_t;
    typedef unsigned long long uacc_t;
    static constexpr double kRange = 4611686018427387904.0;      // 2^62
    __device__ static acc_t round(double x) { return __double2ll_rn(x); }
    __device__ static bool finite(double x)
    {
        return ((unsigned long long)__double_as_longlong(x) & 0x7ff0000000000000ULL) != 0x7ff0000000000000ULL;
    }
};

__device__ __forceinline__ double wave_sum(double f)
{
    for (int m = 32; m >= 1; m >>= 1)
        f += __shfl_xor(f, m);
    return f;
}

template <typename T, int ORDER, int TX>
__global__ __launch_bounds__(kBlock, sizeof(T) == 8 ? 2 : 3) void deform_tile3_grad_kernel(const GridGeom g,
                                                                      const IOView v,
                                                                      const TileGeom tg)
{
    constexpr int NT = ORDER + 1;
    typedef typename GradFixed<T>::acc_t acc_t;
    typedef typename GradFixed<T>::uacc_t uacc_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int phase = 0;      // parity of the per-wave |dY| sum slots
    for (int work = blockIdx.x;; work += gridDim.x) {
    StripPos sp;
    if (!strip_position(tg, work, sp))
        return;
    if (work != (int)blockIdx.x)
        __syncthreads();      // the previous work item is done with the LDS
    strip_prologue(g, v, tg, sp, smem);

    const AxTab* tabx = reinterpret_cast<const AxTab*>(smem + kOffTabX);
    int* sred = reinterpret_cast<int*>(smem + kOffRed);
    const double* sQ = reinterpret_cast<const double*>(smem + kOffQ);
    const HotParams* hp = reinterpret_cast<const HotParams*>(smem + kOffHot);
    acc_t* box = reinterpret_cast<acc_t*>(smem + tg.off_ov);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // a tile is 8 (z) x 8 (y) x TX (x) voxels; a thread owns column (yy, xx) and NV z-slices
    constexpr int NV = TX / 4;
    constexpr int ZSTEP = 8 / NV;
    const int xx = tid % TX, yy = (tid / TX) & 7, zq = tid / (TX * 8);
    const int ntile = (sp.ntile * kT + TX - 1) / TX;
    T* dx = reinterpret_cast<T*>(const_cast<char*>(v.in)) + (int64_t)sp.sample * tg.in_bstride;   // accumulated into
    const T* __restrict__ dy = reinterpret_cast<const T*>(v.out) + (int64_t)sp.sample * tg.out_bstride;

    for (int ti = 0; ti < ntile; ++ti) {
        const int o0[3] = {sp.tz * kT, sp.ty * kT, sp.tx0 * kT + ti * TX};
        int* red = sred + (ti % 3) * 8;

        int start[NV][3];
        T frac[NV][3];
        bool active[NV];
        int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
        int ooff[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int zi = zq + ZSTEP * i;
            const int o[3] = {o0[0] + zi, o0[1] + yy, o0[2] + xx};
            const bool valid = o[0] < tg.out_len[0] && o[1] < tg.out_len[1] && o[2] < tg.out_len[2];
            const bool cst = voxel_coords<T, ORDER>(tg, hp, sQ, tabx[ti * TX + xx], zi, yy, o,
                                                        start[i], frac[i]);
            active[i] = valid && !cst;       // constant-mapped voxels contribute nothing (:928)
            ooff[i] = o[0] * tg.out_stride[0] + o[1] * tg.out_stride[1] + o[2] * tg.out_stride[2];
            if (active[i]) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    lo[h] = min(lo[h], start[i][h]);
                    hi[h] = max(hi[h], start[i][h] + ORDER);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int l = wave_min(lo[h]);
            const int u = wave_max(hi[h]);
            if (lane == 0) {
                atomicMin(&red[h], l);
                atomicMax(&red[3 + h], u);
            }
        }
        __syncthreads();   // B1: box known; the previous tile's flush is done
        const int b0[3] = {red[0], red[1], red[2]};
        const int ext[3] = {red[3] - red[0] + 1, red[4] - red[1] + 1, red[5] - red[2] + 1};
        const bool any = red[3] >= red[0];
        if (tid < 6)       // tile ti + 2's buffer: see the forward kernel
            sred[((ti + 2) % 3) * 8 + tid] = tid < 3 ? 0x7fffffff : (int)0x80000000;
        if (!any)
            continue;      // nothing to scatter (uniform)
        const int pitch = ext[2] <= 8 ? 8 : (ext[2] <= 24 ? 24 : (ext[2] <= 40 ? 40 : (ext[2] <= 56 ? 56 : 0)));
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const int nbox = nrows * pitch;
        const bool fits = pitch > 0 && nbox <= tg.box_cap;
        if (!fits) {
            if (tid < TX / kT && sp.tx0 + ti * (TX / kT) + tid < tg.tiles[2]) {
                const int slot = atomicAdd(&tg.spill[0], 1);
                tg.spill[1 + slot] = sp.sample * tg.ntiles +
                                     (sp.tz * tg.tiles[1] + sp.ty) * tg.tiles[2] + sp.tx0 +
                                     ti * (TX / kT) + tid;
            }
            continue;
        }
        const bool x_inside = b0[2] >= 0 && b0[2] + ext[2] <= tg.in_len[2];
        const float inv_by = 1.0f / (float)by;

        for (int64_t ss = 0; ss < v.nsteps; ++ss, ++phase) {
            int64_t in_off, out_off;
            step_offsets(hp, ss, in_off, out_off);
            if (ss > 0)
                __syncthreads();         // previous step's flush is done with the box
            // zero the accumulators; tile maximum of |dY|
            {
                int* zb = reinterpret_cast<int*>(box);
                const int nz = nbox * (int)(sizeof(acc_t) / sizeof(int));
                for (int e = tid * 4; e < nz; e += kBlock * 4)
                    *reinterpret_cast<int4*>(zb + e) = make_int4(0, 0, 0, 0);
            }
            T gval[NV];
            T gm = 0;
            T* dst = dx + in_off;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                gval[i] = active[i] ? dy[out_off + ooff[i]] : (T)0;
                if (!GradFixed<T>::finite(gval[i])) {
                    // inf / NaN gradient: no fixed-point scale exists -- this voxel scatters its
                    // taps with float atomics straight to global memory (rare, rolled loops)
                    T w0[NT], w1[NT], w2[NT];
                    weights_from_frac<T, ORDER>(frac[i][0], w0);
                    weights_from_frac<T, ORDER>(frac[i][1], w1);
                    weights_from_frac<T, ORDER>(frac[i][2], w2);
#pragma unroll 1
                    for (int t = 0; t < NT * NT * NT; ++t) {
                        const int l0 = t / (NT * NT), l1 = (t / NT) % NT, l2 = t % NT;
                        const int zs = mirror_i32(start[i][0] + l0, tg.in_len[0]);
                        const int ys = mirror_i32(start[i][1] + l1, tg.in_len[1]);
                        const int xs = mirror_i32(start[i][2] + l2, tg.in_len[2]);
                        T wp = w0[0], wq = w1[0], wr = w2[0];
#pragma unroll
                        for (int l = 1; l < NT; ++l) {
                            wp = l0 == l ? w0[l] : wp;
                            wq = l1 == l ? w1[l] : wq;
                            wr = l2 == l ? w2[l] : wr;
                        }
                        unsafeAtomicAdd(dst + (zs * tg.in_stride[0] + ys * tg.in_stride[1] +
                                               xs * tg.in_stride[2]),
                                        gval[i] * wp * wq * wr);
                    }
                    gval[i] = 0;
                }
                gm += fabs(gval[i]);
            }
            // sum of |dY| over the tile: wave reduce, one slot per wave, combined after the barrier
            gm = wave_sum(gm);
            // per-wave sums: float32 in the fixed slot, float64 behind the box (the slot is 32 bytes)
            T* gsum = (sizeof(T) == 4 ? reinterpret_cast<T*>(smem + kOffSum)
                                      : reinterpret_cast<T*>(smem + tg.off_ov + (size_t)tg.box_cap * sizeof(acc_t))) +
                      (phase & 1) * 4;
            if (lane == 0)
                gsum[wave] = gm;
            __syncthreads();             // B2: box zeroed, sum known
            const T gtot = (gsum[0] + gsum[1]) + (gsum[2] + gsum[3]);
            if (gtot == (T)0)
                continue;                // all-zero gradient tile (uniform)
            // fixed-point scale.  Rigorous bound on what can land in one accumulator:
            //   |sum| <= max tap weight * sum over the tile's voxels of |dY|   (+ 1/2 per rounding)
            // so scale = (2^31 - 2^10) / (wmax * sum|dY|) cannot overflow an int32; for white-noise
            // dY that is a resolution of ~3e-8 of max|dY| per contribution -- the level of the
            // float32 rounding in the reference's own `+=` (deform.c:309-312).
            constexpr double kWmax = ORDER == 1 ? 1.0 : ORDER == 2 ? 0.4219 : (ORDER == 3 ? 0.2963
                                    : (ORDER == 4 ? 0.2150 : 0.1664));
            const T scale = (T)(GradFixed<T>::kRange / (kWmax * 1.001 * (double)gtot));
            const T inv_scale = (T)1 / scale;

            // two voxels per trip (four unrolled 64-tap scatters do not fit the register budget: fully
            // unrolled the kernel spills and runs 7 % slower, fully rolled 1 % slower); the voxel's
            // state is picked out of the register arrays by select chains
#pragma unroll 2
            for (int i = 0; i < NV; ++i) {
                int st0 = start[0][0], st1 = start[0][1], st2 = start[0][2];
                T f0 = frac[0][0], f1 = frac[0][1], f2 = frac[0][2], gv = gval[0];
                bool act = active[0];
#pragma unroll
                for (int k = 1; k < NV; ++k) {
                    const bool sel = i == k;
                    st0 = sel ? start[k][0] : st0;
                    st1 = sel ? start[k][1] : st1;
                    st2 = sel ? start[k][2] : st2;
                    f0 = sel ? frac[k][0] : f0;
                    f1 = sel ? frac[k][1] : f1;
                    f2 = sel ? frac[k][2] : f2;
                    gv = sel ? gval[k] : gv;
                    act = sel ? active[k] : act;
                }
                if (!act || gv == (T)0 || ED_DBG(tg.dbg, 128))
                    continue;
                T w0[NT], w1[NT], w2[NT];
                weights_from_frac<T, ORDER>(f0, w0);
                weights_from_frac<T, ORDER>(f1, w1);
                weights_from_frac<T, ORDER>(f2, w2);
                const int rz = st0 - b0[0], ry = st1 - b0[1], rx = st2 - b0[2];
                acc_t* bp = box + (rz * by + ry) * pitch + rx;
                const T gs = gv * scale;
#pragma unroll
                for (int l0 = 0; l0 < NT; ++l0) {
                    const T g0 = gs * w0[l0];
#pragma unroll
                    for (int l1 = 0; l1 < NT; ++l1) {
                        const T g1 = g0 * w1[l1];
                        acc_t* rp = bp + (l0 * by + l1) * pitch;
#pragma unroll
                        for (int l2 = 0; l2 < NT; ++l2)
                            atomicAdd(reinterpret_cast<uacc_t*>(rp + l2), (uacc_t)GradFixed<T>::round(g1 * w2[l2]));
                    }
                }
            }
            __syncthreads();             // B3: all contributions are in
            // flush: one float atomic per touched source element, mirror-mapped (deform.c:791-813)
            // consecutive lanes walk consecutive elements of the padded rows: a row's touched
            // elements are one contiguous run of float atomics
            // (only the ext[2] live cells of each padded row are visited)
            const float inv_ex = 1.0f / (float)ext[2];
            const int ncell = nrows * ext[2];
            for (int e = tid; e < (ED_DBG(tg.dbg, 64) ? 0 : ncell); e += kBlock) {
                const int r = (int)(((float)e + 0.5f) * inv_ex), xi = e - r * ext[2];
                const acc_t acc = box[r * pitch + xi];
                if (acc != 0) {
                    const int zr = (int)(((float)r + 0.5f) * inv_by), yr = r - zr * by;
                    const int zs = mirror_i32(b0[0] + zr, tg.in_len[0]);
                    const int ys = mirror_i32(b0[1] + yr, tg.in_len[1]);
                    const int xs = x_inside ? b0[2] + xi : mirror_i32(b0[2] + xi, tg.in_len[2]);
                    unsafeAtomicAdd(dst + (zs * tg.in_stride[0] + ys * tg.in_stride[1] +
                                           xs * tg.in_stride[2]),
                                    (T)acc * inv_scale);
                }
            }
        }
    }
    }   // work items
}

// ================================================================================================
// direct kernel: no LDS staging.  One block per 8^3 tile, two voxels per thread; displacement from
// the per-call tables (Q, XT: 12 fp64 FMAs per voxel), taps straight from / to global memory with
// the per-axis mirror map of deform.c:791-813.  Two uses:
//   WORKLIST = true : finishes the tiles the LDS kernels could not hold (strong folding, 'wrap'
//                     seams) from the spill worklist;
//   WORKLIST = false: the whole volume, for spline order 0 (one tap per voxel: nothing to stage).
//                     Order 1 runs on the LDS kernels: 256^3 float32 forward 0.27 -> 0.20 ms, gradient 2.1 -> 0.27 ms (eight
//                     global float atomics per voxel were the cost).
// ================================================================================================
template <typename T, int ORDER, bool GRAD, bool WORKLIST>
__global__ __launch_bounds__(kBlock) void deform_tile3_direct_kernel(const GridGeom g, const IOView v,
                                                                     const TileGeom tg)
{
    constexpr int NT = ORDER + 1;
    (void)g;
    const int ntile_total = tg.ntiles * tg.nbatch;
    const int nwork = WORKLIST ? tg.spill[0] : ntile_total;
    const int tid = threadIdx.x;
    const int xx = tid & 7, yy = (tid >> 3) & 7, zq = tid >> 6;
    for (int s = blockIdx.x; s < nwork; s += gridDim.x) {
        int t = WORKLIST ? tg.spill[1 + s] : s;
        const int sample = t / tg.ntiles;
        t -= sample * tg.ntiles;
        T* inp = reinterpret_cast<T*>(const_cast<char*>(v.in)) + (int64_t)sample * tg.in_bstride;
        T* outp = reinterpret_cast<T*>(v.out) + (int64_t)sample * tg.out_bstride;
        const double* qs = tg.q_global + (int64_t)sample * tg.q_bstride;
        const int tx = t % tg.tiles[2];
        t /= tg.tiles[2];
        const int ty = t % tg.tiles[1];
        const int tz = t / tg.tiles[1];
        const int ox = tx * kT + xx, oy = ty * kT + yy;
        if (ox >= tg.out_len[2] || oy >= tg.out_len[1])
            continue;
        const AxTab tx_ = tg.xt_global[ox];
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            const int oz = tz * kT + zq + 4 * i;
            if (oz >= tg.out_len[0])
                continue;
            const int o[3] = {oz, oy, ox};
            const int64_t qrow_id = tg.q_win ? ((int64_t)oz * tg.out_len[1] + oy) * tg.q_strips + (tx * kT) / tg.q_strip_vox
                                             : (int64_t)oz * tg.out_len[1] + oy;
            const double* qrow0 = qs + qrow_id * 4 * tg.ncpx;
            // coordinates (same arithmetic as voxel_coords, tables read from global memory)
            int ci[3];
            T fr[3];
            bool cst = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const double* qrow = qrow0 + h;
                double d = 0.0;
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    d = fma(tx_.w[l], qrow[tx_.idx[l]], d);
                double c = d;
                int bi = o[h] + tg.off[h];
                if (tg.has_affine) {
                    double b = tg.affine[h * 4 + 3];
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                        b = fma(tg.affine[h * 4 + l], (double)o[l], b);
                    c = b + (double)tg.off[h] + d;
                    bi = 0;
                }
                if (!coord_axis_fast<ORDER, T>(c, bi, tg.in_len[h], ci[h], fr[h]))
                    cst = coord_axis_mapped<ORDER, T>((double)bi + c, tg.in_len[h], tg.mode, tg.period[h],
                                                      tg.inv_period[h], ci[h], fr[h]) || cst;
            }
            int tap[3][NT];
            T w[3][NT];
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const int st = cst ? 0 : ci[h] - ORDER / 2;
                weights_from_frac<T, ORDER>(fr[h], w[h]);
#pragma unroll
                for (int l = 0; l < NT; ++l)
                    tap[h][l] = mirror_i32(st + l, tg.in_len[h]) * tg.in_stride[h];
            }
            const int obase = o[0] * tg.out_stride[0] + o[1] * tg.out_stride[1] + o[2] * tg.out_stride[2];
            for (int64_t ss = 0; ss < v.nsteps; ++ss) {
                int64_t in_off, out_off;
                step_offsets(v, ss, in_off, out_off);
                T* src = inp + in_off;
                if (!GRAD) {
                    T val;
                    if (cst) {
                        val = (T)v.cval;
                    } else {
                        T a0 = 0;
#pragma unroll
                        for (int l0 = 0; l0 < NT; ++l0) {
                            T a1 = 0;
#pragma unroll
                            for (int l1 = 0; l1 < NT; ++l1) {
                                const T* p1 = src + (tap[0][l0] + tap[1][l1]);
                                T a2 = 0;
#pragma unroll
                                for (int l2 = 0; l2 < NT; ++l2)
                                    a2 = tap_fma(w[2][l2], p1[tap[2][l2]], a2);
                                a1 = tap_fma(w[1][l1], a2, a1);
                            }
                            a0 = tap_fma(w[0][l0], a1, a0);
                        }
                        val = a0;
                    }
                    __builtin_nontemporal_store(val, outp + (out_off + obase));   // see K1's output store
                } else if (!cst) {
                    const T grad = outp[out_off + obase];
#pragma unroll
                    for (int l0 = 0; l0 < NT; ++l0) {
                        const T g0 = grad * w[0][l0];
#pragma unroll
                        for (int l1 = 0; l1 < NT; ++l1) {
                            T* p1 = src + (tap[0][l0] + tap[1][l1]);
                            const T g1 = g0 * w[1][l1];
#pragma unroll
                            for (int l2 = 0; l2 < NT; ++l2)
                                unsafeAtomicAdd(p1 + tap[2][l2], g1 * w[2][l2]);
                        }
                    }
                }
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------

inline int ceil_log2(int64_t n)
{
    int l = 0;
    while (((int64_t)1 << l) < n)
        ++l;
    return l;
}

inline size_t q_bytes(const GridGeom& g) { return 8 * (size_t)(kT * kT * 4) * (size_t)g.ncp[2]; }
// near-tie list of the label kernel: up to 1M voxel ids (4 MiB), never more than the output has
inline size_t label_list_bytes(const GridGeom& g)
{
    int64_t nvox = 1;
    for (int k = 0; k < 3; ++k)
        nvox *= g.out_len[k];
    const int64_t cap = nvox < (1 << 20) ? nvox : (1 << 20);
    return (size_t)(cap + 32) * sizeof(int) + 64;
}

// the tile kernels keep a strip's 64 Q rows in LDS next to a 32 KiB box: up to this many control columns
inline bool wide_grid(const GridGeom& g) { return kOffQ + q_bytes(g) + 16 + 32768 > (size_t)64 * 1024; }
// wide grids: strips of kWideStripTiles tiles, the columns such a strip touches (conservative; <= kWideMaxWin or the
// row kernel of deform_fast.hip takes the call)
constexpr int kWideStripTiles = 4;
constexpr int kWideMaxWin = 16;
inline int wide_window(const GridGeom& g)
{
    const double r = g.in_len[2] > 1 ? (double)(g.ncp[2] - 1) / (double)(g.in_len[2] - 1) : 0.0;
    return (int)std::floor(kWideStripTiles * kT * r) + 6;
}
// x table + the per-sample margins of K1's sampled boxes (TileGeom::slack) behind it
inline size_t xt_only_bytes(const GridGeom& g) { return (sizeof(AxTab) * (size_t)g.out_len[2] + 63) & ~(size_t)63; }
inline size_t xt_block_bytes(const GridGeom& g, int nbatch) { return xt_only_bytes(g) + (((size_t)nbatch * 32 + 63) & ~(size_t)63); }
// float64 volumes (round 5): the per-strip layout also where the whole grid would still fit next to a box -- with 9 to 14
// control columns the Q rows take 18-29 KB of LDS, the level-2 kernel is left with half a box of 8-byte cells and a
// rough deformation sends most tiles to the direct kernel (256^3, 13^3 grid, order 3 gradient: 8.0 ms, of which 6.8 in
// float64 global atomics; per-strip tables of 7 columns: see profiles/r05_bench_misc.txt).  Worth it when the window is
// narrower than the grid.
inline bool wide_optional(const GridGeom& g)
{
    if (wide_grid(g))
        return false;
    const int win = wide_window(g);
    const int64_t strips = (g.out_len[2] + kWideStripTiles * kT - 1) / (kWideStripTiles * kT);
    if (win + 2 > g.ncp[2] || win > kWideMaxWin || strips > 256)
        return false;
    return 8.0 * (double)g.out_len[0] * (double)g.out_len[1] * 4.0 * (double)(strips * win) <= (double)((size_t)512 << 20);
}
inline size_t q_global_bytes(const GridGeom& g, bool wide_opt = false)
{
    size_t cols = (size_t)g.ncp[2];
    if (wide_grid(g) || (wide_opt && wide_optional(g))) {
        // Q[o_z][o_y][strip][window][4] (TileGeom::q_win)
        const size_t strips = (size_t)((g.out_len[2] + kWideStripTiles * kT - 1) / (kWideStripTiles * kT));
        const size_t per_strip = strips * (size_t)wide_window(g);
        cols = per_strip > cols ? per_strip : cols;
    }
    return 8 * (size_t)g.out_len[0] * (size_t)g.out_len[1] * 4 * cols;
}
// The geometry buffer of the forward route of deform_k1z.hip (SideLane::geo_ptr, its own allocation):
//   counters (4 KiB, cleared at allocation) | ZGen (512) | z table | tile records (32 bytes per tile), flags and summaries
//   (4 + 4 per strip, at most one strip per tile), two work lists dealt per XCD (entry k of XCD x at slot 8 k + x: 32 + 32
//   per strip, in case one XCD gets them all), records of the tiles' z halves (64) | step offsets | R
constexpr size_t kK1zMaxSteps = 4096;
inline size_t k1z_zt_bytes(const GridGeom& g) { return (sizeof(AxTab) * (size_t)g.out_len[0] + 63) & ~(size_t)63; }
inline size_t k1z_geo_bytes(const GridGeom& g, int64_t ntiles, int nbatch)
{
    return 4096 + 512 + k1z_zt_bytes(g) + (((size_t)ntiles * (size_t)nbatch * 168 + 63) & ~(size_t)63) + kK1zMaxSteps * 16 +
           ((k1z_r_bytes(g) * (size_t)nbatch + 63) & ~(size_t)63);
}

// ================================================================================================
// label kernel: order 0, any dtype, input dtype == output dtype -- a resampled label map is a
// COPY of source elements, so only the choice of the source index has to match the reference.
// Coordinates come from the per-call tables like everywhere on the fast path (12 fp64 FMAs per
// voxel); a voxel whose coordinate lies within 1e-6 of a decision boundary -- a half-integer (the
// rounding floor(c + 0.5)), or an integer once it is outside the array (the break points of the
// boundary maps, the constant test) -- is re-evaluated in the reference's own evaluation order
// (ed_exact_coord.h, FMA contraction off), which is bit-comparable.  The fast coordinate is within
// ~1e-11 of that, so every other voxel decides identically: the output is bit-equal to the exact
// kernel's at ~20x its speed (a few voxels per million take the slow path).
// ================================================================================================
template <typename W>
__global__ __launch_bounds__(kBlock) void deform_tile3_label_kernel(const GridGeom g, const IOView v,
                                                                    const TileGeom tg)
{
    const W* inp = reinterpret_cast<const W*>(v.in);
    W* outp = reinterpret_cast<W*>(v.out);
    const int ntile_total = tg.tiles[0] * tg.tiles[1] * tg.tiles[2];
    const int tid = threadIdx.x;
    const int xx = tid & 7, yy = (tid >> 3) & 7, zq = tid >> 6;
    constexpr double kEps = 1e-6;
    for (int s = blockIdx.x; s < ntile_total; s += gridDim.x) {
        int t = s;
        const int tx = t % tg.tiles[2];
        t /= tg.tiles[2];
        const int ty = t % tg.tiles[1];
        const int tz = t / tg.tiles[1];
        const int ox = tx * kT + xx, oy = ty * kT + yy;
        if (ox >= tg.out_len[2] || oy >= tg.out_len[1])
            continue;
        const AxTab tx_ = tg.xt_global[ox];
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            const int oz = tz * kT + zq + 4 * i;
            if (oz >= tg.out_len[0])
                continue;
            const int o[3] = {oz, oy, ox};
            const double* qrow0 = tg.q_global + ((int64_t)oz * tg.out_len[1] + oy) * 4 * tg.ncpx;
            double raw[3];
            bool tie = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const double* qrow = qrow0 + h;
                double d = 0.0;
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    d = fma(tx_.w[l], qrow[tx_.idx[l]], d);
                double b;
                if (tg.has_affine) {
                    b = tg.affine[h * 4 + 3];
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                        b = fma(tg.affine[h * 4 + l], (double)o[l], b);
                    b += (double)tg.off[h];
                } else {
                    b = (double)(o[h] + tg.off[h]);
                }
                raw[h] = b + d;
                const double fr = raw[h] - floor(raw[h]);
                const bool outside = !(raw[h] >= kEps && raw[h] <= (double)(tg.in_len[h] - 1) - kEps);
                tie = tie || fabs(fr - 0.5) < kEps || (outside && (fr < kEps || fr > 1.0 - kEps)) ||
                      !(raw[h] == raw[h]);
            }
            if (tie) {
                // left to the tie kernel (the reference's own arithmetic for this voxel)
                const int slot = atomicAdd(&tg.label_list[0], 1);
                if (slot < tg.label_cap)
                    tg.label_list[1 + slot] = (oz * tg.out_len[1] + oy) * tg.out_len[2] + ox;
                continue;
            }
            int src_idx = 0;
            bool cst = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const double c = map_coordinate_fast(raw[h], tg.in_len[h], tg.mode, tg.period[h],
                                                     tg.inv_period[h]);
                cst = cst || !(c > -1.0);
                const int st = (int)floor(c + 0.5);
                src_idx += mirror_i32(cst ? 0 : st, tg.in_len[h]) * tg.in_stride[h];
            }
            const int obase = o[0] * tg.out_stride[0] + o[1] * tg.out_stride[1] + o[2] * tg.out_stride[2];
            for (int64_t ss = 0; ss < v.nsteps; ++ss) {
                int64_t in_off, out_off;
                step_offsets(v, ss, in_off, out_off);
                if (cst)
                    store_forward(reinterpret_cast<char*>(outp + (out_off + obase)), v.out_dtype, v.cval);
                else if (sizeof(W) == 8)
                    // 64-bit integers: the reference takes every value through a double and the
                    // rounding / clamping store (deform.c:863-887,906-919), which changes labels
                    // beyond 2^53 and near the type's limits -- reproduce that round trip
                    store_forward(reinterpret_cast<char*>(outp + (out_off + obase)), v.out_dtype,
                                  load_as_double(reinterpret_cast<const char*>(inp + (in_off + src_idx)),
                                                 v.in_dtype));
                else
                    __builtin_nontemporal_store(inp[in_off + src_idx], outp + (out_off + obase));
            }
        }
    }
}

// edhip_profile_*: HIP events around the level-1 launch only (the dominant kernel of a call),
// recorded on the stream the kernel is launched on.  Off unless bench.py asks for it.
std::atomic<int> g_profile{0};
thread_local hipEvent_t t_ev0 = nullptr, t_ev1 = nullptr;
thread_local bool t_ev_pending = false;

void profile_mark(bool after, hipStream_t stream)
{
    if (!g_profile.load(std::memory_order_relaxed))
        return;
    if (!t_ev0) {
        if (hipEventCreate(&t_ev0) != hipSuccess || hipEventCreate(&t_ev1) != hipSuccess) {
            t_ev0 = t_ev1 = nullptr;
            return;
        }
    }
    (void)hipEventRecord(after ? t_ev1 : t_ev0, stream);
    if (after)
        t_ev_pending = true;
}

template <typename T, int ORDER, bool PAIR, bool GRAD>
hipError_t launch_tile(const GridGeom& g, const IOView& v, hipStream_t stream, const DeformBatch* batch)
{
    TileGeom tg;
    IOView ve = v;
    const int nb = batch ? batch->nbatch : 1;
    tg.nbatch = nb;
    tg.in_bstride = batch ? batch->in_bstride / (int64_t)sizeof(T) : 0;
    tg.out_bstride = batch ? batch->out_bstride / (int64_t)sizeof(T) : 0;
    tg.disp_bstride = batch ? batch->disp_bstride : 0;
    for (int k = 0; k < 3; ++k) {
        tg.in_len[k] = (int)g.in_len[k];
        tg.out_len[k] = (int)g.out_len[k];
        tg.tiles[k] = (int)((g.out_len[k] + kT - 1) / kT);
        tg.in_stride[k] = (int)(v.in_stride[k] / (int64_t)sizeof(T));
        tg.out_stride[k] = (int)(v.out_stride[k] / (int64_t)sizeof(T));
        tg.off[k] = (int)g.off[k];
        // boundary-map period of the axis (deform.c:56,65,75 / :94,104,114)
        const double len = (double)g.in_len[k];
        tg.period[k] = v.mode == EDHIP_MODE_MIRROR ? 2 * len - 2
                       : (v.mode == EDHIP_MODE_REFLECT ? 2 * len : len - 1);
        tg.inv_period[k] = tg.period[k] > 0 ? 1.0 / tg.period[k] : 0.0;
    }
    tg.mode = v.mode;
    tg.has_affine = g.has_affine;
    for (int k = 0; k < 12; ++k)
        tg.affine[k] = g.affine[k];
    for (int l = 0; l < v.nstep; ++l) {
        ve.in_step_stride[l] = v.in_step_stride[l] / (int64_t)sizeof(T);
        ve.out_step_stride[l] = v.out_step_stride[l] / (int64_t)sizeof(T);
    }
    // strips of 8 tiles amortise the prologue; small outputs (crops) take shorter strips so that the
    // launch still has a few workgroups per CU (even lengths for the gradient: its kernel walks 16-wide
    // tiles; the forward kernels go down to one tile per workgroup -- a tile is a serial chain of
    // coordinates, box reduction, staging and gather, ~8 us, and a 32^3 volume has 64 of them)
    // Wide control grid: the level-1 kernels of deform_hot.hip on per-strip Q tables, in self-serve form, or nothing
    // (hipErrorNotSupported before anything is launched: the row kernel of deform_fast.hip then takes the call --
    // at 24.7 ms for the gradient of a 256^3 volume with a 16^3 grid, where this route takes 0.x ms)
    const bool wide_opt = sizeof(T) == 8 && std::is_floating_point<T>::value && ORDER >= 1 && nb == 1 &&
                          !ed_env("EDHIP_RECORDS") && !ed_env("EDHIP_NO_WIDE64") && wide_optional(g);
    const bool wide = wide_grid(g) || wide_opt;
    tg.q_win = tg.q_strip_vox = 0;
    tg.q_strips = 1;
    // (orders 4 / 5: the one-wave kernels of deform_wave.hip read their Q columns from global memory as they walk --
    // no LDS limit on the grid's width: plain tables, and what they cannot hold goes straight to the direct level,
    // because the level-2 kernel stages Q rows like the others.  256^3, 16^3 grid, order 5 on the row kernel:
    // forward 4.7 ms, gradient 83 ms)
    constexpr bool kWaveOrder = std::is_same<T, float>::value && ORDER >= 4;
    const bool wide_wave = wide && kWaveOrder && tg.in_stride[2] == 1 && tg.out_stride[2] == 1 && !ed_env("EDHIP_NO_HOT") &&
                           !ed_env("EDHIP_WAVE");
    // (round 5: the general kernels read the per-strip layout too -- float64 volumes and layouts without unit stride along x
    // no longer fall to the row kernel: 256^3 float64, 16^3 grid, order 3 gradient 22.3 ms -> see profiles/r05_bench_misc.txt)
    bool wide_hot = false;       // the level-1 kernels of deform_hot.hip / deform_k1.hip / deform_wave.hip take the call
    if (wide) {
        if constexpr (!(std::is_floating_point<T>::value && ORDER >= 1))
            return hipErrorNotSupported;
        if (nb != 1 || ed_env("EDHIP_RECORDS"))
            return hipErrorNotSupported;
        wide_hot = std::is_same<T, float>::value && tg.in_stride[2] == 1 && tg.out_stride[2] == 1 && !ed_env("EDHIP_NO_HOT") &&
                   !ed_env("EDHIP_WAVE");
        if (!wide_wave) {
            const int win = wide_window(g);
            const int64_t strips = (g.out_len[2] + kWideStripTiles * kT - 1) / (kWideStripTiles * kT);
            if (win > kWideMaxWin || strips > 256)
                return hipErrorNotSupported;
            tg.q_win = win;
            tg.q_strip_vox = kWideStripTiles * kT;
            tg.q_strips = (int)strips;
        }
    }
    if (v.out16) {
        // 16-bit output side: the level-1 kernels of deform_hot.hip (orders 1-3) in self-serve form, or nothing
        if constexpr (!(std::is_same<T, float>::value && ORDER >= 1 && ORDER <= 3))
            return hipErrorNotSupported;
        if (nb != 1 || tg.in_stride[2] != 1 || tg.out_stride[2] != 1 || ed_env("EDHIP_NO_HOT") || ed_env("EDHIP_WAVE") ||
            ed_env("EDHIP_RECORDS"))
            return hipErrorNotSupported;
    }
    // (the forward kernels prefer strips of 4: 256^3 order 3, K1 call 237.5 -> 228.7 us, sigma 10 342.8 -> 338.0, order 1
    // 159.4 -> 155.8 -- twice the workgroups for the tail of the launch to be dealt from; the gradient kernel, which
    // walks a strip in 16-wide tiles, prefers 8: 310.7 against 313.3 us; profiles/r04_bench_misc.txt)
    tg.strip_tiles = GRAD ? kStrip : kStrip / 2;
    if (tg.q_win)
        tg.strip_tiles = kWideStripTiles;       // (the per-strip Q layout is made for exactly this length)
    while (!tg.q_win && tg.strip_tiles > (GRAD ? 2 : 1) &&
           (int64_t)nb * tg.tiles[0] * tg.tiles[1] * ((tg.tiles[2] + tg.strip_tiles - 1) / tg.strip_tiles) < 1024)
        tg.strip_tiles >>= 1;
#ifdef EDHIP_EXPERIMENTS
    if (const char* st = ed_env("EDHIP_STRIP"))
        tg.strip_tiles = atoi(st) >= 1 && atoi(st) <= kStrip ? atoi(st) : tg.strip_tiles;
#endif
    tg.strips_x = (tg.tiles[2] + tg.strip_tiles - 1) / tg.strip_tiles;
    const int64_t nstrips = (int64_t)tg.tiles[0] * tg.tiles[1] * tg.strips_x;
    const int64_t ntiles = (int64_t)tg.tiles[0] * tg.tiles[1] * tg.tiles[2];
    if (nstrips <= 0)
        return hipSuccess;
    if (ntiles * nb > 0x3fffffffLL || nstrips * nb > 0x3fffffffLL)
        return hipErrorInvalidValue;
    tg.nstrips = (int)nstrips;
    tg.ntiles = (int)ntiles;
    tg.q_bstride = (long long)(q_global_bytes(g, wide_opt) / 8);
    // (wide grids: the kernels see a grid of q_win columns, the strip's window)
    const size_t qcols = tg.q_win ? (size_t)tg.q_win : (size_t)g.ncp[2];
    tg.ncpx = (int)qcols;
    // LDS: head | Q | overlay (box | D, P).  K1 float32 odd orders: two shifted copies of 4096
    // elements; otherwise one copy (6144 x 4 bytes or 4096 x 8 bytes)
    size_t box;
    if (GRAD) {
        // 32 KiB of fixed-point cells (int32 for float32, int64 for float64) + the float64 wave sums
        tg.box_cap = (int)(32768 / sizeof(T));
        box = 32768 + 64;
    } else if (PAIR) {
        tg.box_cap = 3704;      // 57 * 64 + 56; two copies + head + Q stay under 40 KiB -> 4 blocks per CU
        box = 2 * 3704 * sizeof(T);
    } else {
        // single copy: orders 4 / 5 have 5- / 6-wide windows and need the larger budget
        tg.box_cap = sizeof(T) == 4 ? (ORDER >= 4 ? 8192 : 6144) : 4096;
        box = (size_t)tg.box_cap * sizeof(T);
    }
    tg.off_ov = (int)(kOffQ + ((8 * (size_t)(kT * kT * 4) * qcols + 15) & ~(size_t)15));
    if (tg.off_ov + box > (size_t)64 * 1024) {
        // the Q rows of a wide window leave less than the standard box inside the 64 KiB a block may ask for
        const size_t room = (size_t)64 * 1024 - tg.off_ov;
        if (room < (size_t)16 * 1024)
            return hipErrorNotSupported;
        if (GRAD) {
            tg.box_cap = (int)(((room - 64) / sizeof(T)) & ~(size_t)15);
            box = (size_t)tg.box_cap * sizeof(T) + 64;
        } else if (PAIR) {
            size_t cap = room / (2 * sizeof(T));
            cap = ((cap - 56) / 64) * 64 + 56;
            tg.box_cap = (int)cap;
            box = 2 * cap * sizeof(T);
        } else {
            tg.box_cap = (int)((room / sizeof(T)) & ~(size_t)15);
            box = (size_t)tg.box_cap * sizeof(T);
        }
    }
    size_t overlay = (box + 15) & ~(size_t)15;
    size_t lds = tg.off_ov + overlay;
    if (const char* pad = ed_env("EDHIP_LDS_PAD"))
        lds += (size_t)atoi(pad);
    {
        const char* dbg = ed_env("EDHIP_TILE_DBG");
        tg.dbg = dbg ? atoi(dbg) : 0;
    }
    // scratch: first-level spill list | second-level spill list | x table | Q
    const size_t list_bytes = (sizeof(int) * ((size_t)ntiles * nb + 1) + 63) & ~(size_t)63;
    const size_t xt_bytes = xt_block_bytes(g, nb);
    const size_t q_all = ((q_global_bytes(g, wide_opt) * (size_t)nb) + 63) & ~(size_t)63;
    hipError_t e = hipSuccess;
    // (edhip_deform reserved deform_tile_workspace_bytes() up front, so this does not move the
    // prefiltered control grid that may sit in the head of the workspace)
    void* ws = workspace_reserve(stream, kWorkspaceGridBytes + 2 * list_bytes + xt_bytes + q_all +
                                             label_list_bytes(g), &e);
    if (!ws)
        return e;
    ws = (char*)ws + kWorkspaceGridBytes;
    int* list_a = (int*)ws;
    int* list_b = (int*)((char*)ws + list_bytes);
    // (a fixed place in the stream's workspace, whatever the geometry of the call: the unused tail of the
    // control-grid head, in front of edhip_source_box's 64 result bytes)
    tg.hint = (int*)((char*)ws - 128);
    tg.hint_host = nullptr;
    tg.hint_seq = 0;
    tg.xt_global = (const AxTab*)((char*)ws + 2 * list_bytes);
    tg.slack = nullptr;          // (set by the route that reads it: K1 of deform_k1.hip)
    tg.slack_scale = 0.0;
    tg.q_global = (const double*)((char*)ws + 2 * list_bytes + xt_bytes);
    tg.worklist = nullptr;
    tg.spill = list_a;
    tg.spill_next = list_b;
    tg.label_list = nullptr;
    tg.label_cap = 0;
    if (std::is_integral<T>::value) {
        tg.label_list = (int*)((char*)ws + 2 * list_bytes + xt_bytes + q_all);
        tg.label_cap = (int)(label_list_bytes(g) / sizeof(int)) - 32;
    }
    // level-1 spill feedback (float32, the kernels of deform_hot.hip and deform_wave.hip): what the recent
    // calls of this geometry on this stream reported decides between the standard and the large boxes
    SpillHint* sh = nullptr;
    bool large_boxes = false, huge_boxes = false;
    // self-serve: the recent calls of this geometry left at most a handful of tiles to the spill list -- level 1
    // then takes such tiles itself (straight from / to global memory) and the two spill launches are not made
    bool self_serve = false, self_serve_boxes = false;
    if constexpr (std::is_same<T, float>::value && ORDER >= 1) {
        sh = ed_env("EDHIP_NO_SPILL_HINT") ? nullptr : spill_hint(stream);
        if (sh) {
            auto geometry_key = [&](bool grad) {
                unsigned long long key = 1469598103934665603ull;
                auto mix = [&](unsigned long long x) { key = (key ^ x) * 1099511628211ull; };
                for (int k = 0; k < 3; ++k) {
                    mix((unsigned long long)g.in_len[k]);
                    mix((unsigned long long)g.out_len[k]);
                    mix((unsigned long long)g.off[k]);
                    mix((unsigned long long)g.ncp[k]);
                }
                mix((unsigned long long)ORDER | ((unsigned long long)grad << 8) | ((unsigned long long)v.mode << 16) |
                    ((unsigned long long)g.has_affine << 24) | ((unsigned long long)nb << 32));
                return key;
            };
            const unsigned long long key = geometry_key(GRAD);
            sh->absorb();
            // (forward, K1z: the large boxes pay from ~6 % of the tiles beyond the standard box -- 256^3, sigma 7.5: 1.2 %,
            // 202 against 225 us; sigma 10: 8.9 %, 249 against 237 us -- profiles/r06_sigma_sweep.txt)
            large_boxes = sh->fraction(key) > (GRAD ? 0.10f : 0.06f);
            // (K2 counts a tile beyond the LARGE box kHintHuge times: a tenth of the tiles there -> the huge boxes)
            huge_boxes = GRAD && sh->fraction(key) > 0.10f * (float)tile::kHintHuge;
            // Forward: a tile gathered straight from global memory costs its workgroup ~10 us, so a handful may stay.
            // Gradient: 64 global float atomics per voxel, ~50 us per 16-wide tile (31 such tiles of a 128^3 volume
            // took K2 from 56 to 141 us).  A gradient call WITH the forward call's boxes takes its oversize tiles as
            // x-halves, and what is left is what the forward call could not hold either: it serves itself when the
            // FORWARD calls of the geometry left at most a few tiles (decided below, where the boxes are known to be
            // there); without the boxes only where its own recent calls left nothing at all.
            self_serve = sh->known(key) && sh->fraction(key) * (float)(ntiles * nb) <= (GRAD ? 0.5f : 64.f);
            if (GRAD) {
                const unsigned long long fkey = geometry_key(false);
                self_serve_boxes = sh->known(fkey) && sh->fraction(fkey) * (float)(ntiles * nb) <= 8.f;
            }
#ifdef EDHIP_EXPERIMENTS
            if (const char* ss = ed_env("EDHIP_SELF_SERVE"))
                self_serve = self_serve_boxes = atoi(ss) != 0;
            if (const char* lb = ed_env("EDHIP_LARGE_BOXES"))       // 0 never, 1 always, 2 forward only, 3 gradient only
                large_boxes = atoi(lb) == 1 || (atoi(lb) == 2 && !GRAD) || (atoi(lb) == 3 && GRAD) || atoi(lb) == 4;
            if (const char* lb = ed_env("EDHIP_LARGE_BOXES"))
                huge_boxes = GRAD && atoi(lb) == 4;
#endif
            tg.hint_host = sh->dev;
            tg.hint_seq = sh->begin_call(key, (unsigned)(ntiles * nb));
        }
    }
    // (a call without spill feedback -- integer volumes, order 0 -- must leave the counters of the float call in
    // front of it alone: its tables kernel used to reset them unreported, and a multi-input call such as
    // BASELINE cfg4 never learned that its geometry does not spill)
    if (!sh)
        tg.hint = nullptr;
    tg.keep_mode = 0;
    tg.keep_stash = nullptr;
    tg.keep_flags = nullptr;
    // the per-call tables launch; the float32 benchmark route decides about the forward -> gradient
    // hand-over first (the tables kernel keeps / checks the control-grid values for it)
    bool tables_done = false;
    auto launch_tables = [&]() {
        if (tables_done || e != hipSuccess)
            return;
        tables_done = true;
        unsigned fill_blocks = 0;
        tg.zero_ptr = nullptr;
        tg.zero_bytes = 0;
        if (GRAD && batch && batch->zero_ptr && !batch->zero_done && nb == 1 && ((uintptr_t)batch->zero_ptr & 15) == 0) {
            tg.zero_ptr = batch->zero_ptr;
            tg.zero_bytes = batch->zero_bytes;
            const long long want = (batch->zero_bytes + 65535) / 65536;       // 64 KiB per workgroup and round
            fill_blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
            batch->zero_done = true;
        }
        GridPrefilter gp;
        memset(&gp, 0, sizeof(gp));
        // a raw grid is filtered inside this launch when the launch's LDS holds it next to the plane slab (the kernel
        // has ~1.2 KB of static LDS on top: a flat grid of 1342-1365 points would pass 64 KiB); otherwise gridpf_done
        // stays false and edhip_deform issues the one-workgroup prefilter first
        bool own = false;
        if (batch && batch->gridpf && nb == 1 && batch->gridpf->total <= 4096 &&
            sizeof(double) * (3 * (size_t)g.ncp[1] * (size_t)g.ncp[2] + (size_t)batch->gridpf->total) <= 60 * 1024) {
            gp = *batch->gridpf;
            gp.zero_ptr = nullptr;
            gp.zero_bytes = 0;
            own = true;
        }
        hipLaunchKernelGGL(tile_tables_kernel, dim3((unsigned)g.out_len[0] + fill_blocks, (unsigned)nb), dim3(kBlock),
                           sizeof(double) * (3 * (size_t)g.ncp[1] * (size_t)g.ncp[2] + (size_t)gp.total), stream, g, tg,
                           gp);
        e = hipGetLastError();
        if (own && e == hipSuccess)
            batch->gridpf_done = true;       // (only a launch that went out has filtered the grid: ADVICE r5)
    };
    if (std::is_integral<T>::value || ORDER < 1)
        launch_tables();
    if constexpr (std::is_integral<T>::value) {
        if (nb != 1)
            return hipErrorNotSupported;      // the label kernels take one volume per call
        if (e == hipSuccess && v.order >= 1) {
            // 8- / 16-bit integer volumes, orders 1-5: the wave-per-tile kernel with fp64 taps; the voxels it
            // lists (value near a rounding tie, coordinate on a boundary) are redone by the exact kernel
            HotGeom hg;
            memset(&hg, 0, sizeof(hg));
            hg.q = tg.q_global;
            hg.xt = tg.xt_global;
            hg.spill = tg.spill;
            hg.q_bstride = tg.q_bstride;
            for (int k = 0; k < 3; ++k) {
                hg.in_len[k] = tg.in_len[k];
                hg.out_len[k] = tg.out_len[k];
                hg.off[k] = tg.off[k];
                hg.tiles[k] = tg.tiles[k];
                hg.period[k] = tg.period[k];
                hg.inv_period[k] = tg.inv_period[k];
            }
            hg.vol_sz = tg.in_stride[0];
            hg.vol_sy = tg.in_stride[1];
            hg.img_sz = tg.out_stride[0];
            hg.img_sy = tg.out_stride[1];
            hg.ntiles = tg.ntiles;
            hg.ncpx = tg.ncpx;
            hg.mode = tg.mode;
            hg.has_affine = tg.has_affine;
            hg.cval = (float)ve.cval;
            hg.cvald = ve.cval;
            hg.nstep = ve.nstep;
            hg.nsteps = ve.nsteps;
            for (int l = 0; l < ve.nstep; ++l) {
                hg.step_len[l] = ve.step_len[l];
                hg.vol_step[l] = ve.in_step_stride[l];
                hg.img_step[l] = ve.out_step_stride[l];
            }
            for (int k = 0; k < 12; ++k)
                hg.affine[k] = tg.affine[k];
            hg.tie_list = tg.label_list;
            hg.tie_cap = tg.label_cap;
            hg.strip_tiles = 4;
            while (hg.strip_tiles > 1 &&
                   (int64_t)tg.tiles[0] * tg.tiles[1] * ((tg.tiles[2] + hg.strip_tiles - 1) / hg.strip_tiles) < 8192)
                hg.strip_tiles >>= 1;
            hg.strips_x = (tg.tiles[2] + hg.strip_tiles - 1) / hg.strip_tiles;
            const int64_t wstrips = (int64_t)tg.tiles[0] * tg.tiles[1] * hg.strips_x;
            if (wstrips > 0x3fffffffLL)
                return hipErrorNotSupported;
            hg.nstrips = hg.total_strips = (int)wstrips;
            const size_t wlds = wave_lds_bytes(false, 3, &hg.box_cap);
            const unsigned wblk = (unsigned)(((wstrips + 7) / 8) * 8);
            e = launch_wave_int(hg, v.order, v.in_dtype, ve.in, ve.out, wblk, wlds, stream);
            if (e == hipSuccess)
                e = launch_deform_exact_list(g, v, tg.label_list, tg.label_cap, stream);
            return e;
        }
        if (e == hipSuccess) {
            const unsigned nblk = (unsigned)(ntiles < (1 << 20) ? ntiles : (1 << 20));
            hipLaunchKernelGGL(deform_tile3_label_kernel<T>, dim3(nblk), dim3(kBlock), 0, stream, g, ve, tg);
            e = hipGetLastError();
            // the near-tie voxels, in the reference's evaluation order: the exact kernel's list form (control grid in
            // LDS; it leaves at once when the list is empty).  A kernel of this file used to do it with the grid in
            // global memory: its handful of voxels sat in ONE wave that walked 192 dependent loads -- 50 us behind a
            // 29 us label kernel (128^3 uint8).
            if (e == hipSuccess)
                e = launch_deform_exact_list(g, v, tg.label_list, tg.label_cap, stream);
        }
        return e;
    } else {
    if (ORDER < 1) {
        // order 0: one tap per voxel, no source box, straight from / to global memory
        if (e == hipSuccess) {
            const unsigned nblk = (unsigned)(ntiles * nb < (1 << 20) ? ntiles * nb : (1 << 20));
            hipLaunchKernelGGL((deform_tile3_direct_kernel<T, ORDER, GRAD, false>), dim3(nblk),
                               dim3(kBlock), 0, stream, g, ve, tg);
            e = hipGetLastError();
        }
        return e;
    }
    // ---- level 1: strips, small boxes, highest occupancy ------------------------------------------
    bool hot_done = false;
    bool served_all = false;        // level 1 ran in self-serve form: no spill list to work off
    HotGeom hg;
    if (e == hipSuccess) {
        const unsigned nblk = (unsigned)(((nstrips * nb + 7) / 8) * 8);
        [[maybe_unused]] constexpr bool kBenchKernel = PAIR && ORDER == 3 && sizeof(T) == 4;
        if constexpr (std::is_same<T, float>::value && ORDER >= 1) {
            // float32, unit stride along x on both sides: the benchmark kernels of deform_hot.hip
            if (tg.in_stride[2] == 1 && tg.out_stride[2] == 1 && !ED_DBG(tg.dbg, 512) && !ed_env("EDHIP_NO_HOT")) {
                memset(&hg, 0, sizeof(hg));
                hg.vol_r = reinterpret_cast<const float*>(ve.in);
                hg.vol_w = reinterpret_cast<float*>(const_cast<char*>(ve.in));
                hg.img_r = reinterpret_cast<const float*>(ve.out);
                hg.img_w = reinterpret_cast<float*>(ve.out);
                hg.q = tg.q_global;
                hg.xt = tg.xt_global;
                hg.spill = tg.spill;
                hg.vol_bstride = tg.in_bstride;
                hg.img_bstride = tg.out_bstride;
                hg.q_bstride = tg.q_bstride;
                for (int k = 0; k < 3; ++k) {
                    hg.in_len[k] = tg.in_len[k];
                    hg.out_len[k] = tg.out_len[k];
                    hg.off[k] = tg.off[k];
                    hg.tiles[k] = tg.tiles[k];
                    hg.period[k] = tg.period[k];
                    hg.inv_period[k] = tg.inv_period[k];
                }
                hg.vol_sz = tg.in_stride[0];
                hg.vol_sy = tg.in_stride[1];
                hg.img_sz = tg.out_stride[0];
                hg.img_sy = tg.out_stride[1];
                hg.strips_x = tg.strips_x;
                hg.strip_tiles = tg.strip_tiles;
                hg.nstrips = tg.nstrips;
                hg.total_strips = tg.nstrips * nb;
                hg.ntiles = tg.ntiles;
                const int hot_cols = tg.q_win ? tg.q_win : tg.ncpx;       // (wide grids: the strip's window of columns)
                hg.ncpx = hot_cols;
                hg.q_strips = tg.q_strips;
                hg.mode = tg.mode;
                hg.has_affine = tg.has_affine;
                hg.dbg = tg.dbg;
                hg.self_serve = (self_serve && ORDER <= 3) ? 1 : 0;      // (orders 4 / 5: the one-wave kernels, which spill as before)
                hg.io16 = v.out16;
                if (v.out16 || wide)
                    hg.self_serve = 1;          // (the spill levels know neither 16-bit storage nor per-strip tables)
                hg.cval = (float)ve.cval;
                hg.nstep = ve.nstep;
                hg.nsteps = ve.nsteps;
                for (int l = 0; l < ve.nstep; ++l) {
                    hg.step_len[l] = ve.step_len[l];
                    hg.vol_step[l] = ve.in_step_stride[l];
                    hg.img_step[l] = ve.out_step_stride[l];
                }
                for (int k = 0; k < 12; ++k)
                    hg.affine[k] = tg.affine[k];
                size_t hlds = 0;
                // forward, orders 1-3: K1 of round 5 (deform_k1.hip: sampled tile boxes, fast tiles); the profiling
                // build can still run the kernel it replaced (EDHIP_K1_OLD)
                [[maybe_unused]] const bool k1_new = !GRAD && ORDER <= 3 && tg.strip_tiles <= 8 && !ed_env("EDHIP_K1_OLD") &&
                                                     !ed_env("EDHIP_RECORDS");
                // round 6: strips along z on R tables (deform_k1z.hip) for every grid its geometry kernel holds in LDS
#ifdef EDHIP_NO_K1Z          // (A/B build of the shipped library without the z-walk route: tools/abl_k1_size.sh)
                const bool k1z = false;
#else
                // Which forward kernel (a function of the call's arguments alone: the two routes differ in the last bits).
                // Measured on shipped builds (profiles/r06_k1_route_sweep.txt): on mild fields the z-walk wins where the
                // x-strip kernel stumbles -- single volumes whose z and y extents are multiples of 256 (256^3 +7 %,
                // 512 x 256 x 256 +18 %) -- and loses elsewhere (128^3 -50 %, 192^3 -24 %, 288^3 -10 %, batches -7 ... -26 %: its
                // geometry kernel costs 20-35 us per call and its tables are per sample); on strong fields (sigma 15) it wins
                // everywhere by 5-35 %.  The library cannot see the field's strength without a host round trip: the caller
                // can say so (EDHIP_FLAG_STRONG_FIELD).
                const bool z_shape = nb == 1 && g.out_len[0] % 256 == 0 && g.out_len[1] % 256 == 0 && g.in_len[0] % 256 == 0 &&
                                     g.in_len[1] % 256 == 0;
                const bool z_pick = ed_env("EDHIP_K1Z_ALWAYS") || (batch && batch->strong) || z_shape;
                const bool k1z = k1_new && !wide && z_pick && k1z_supported(g) && ve.nsteps <= (int64_t)kK1zMaxSteps && !ed_env("EDHIP_K1_R5");
#endif
                ZGeom zg;
                memset(&zg, 0, sizeof(zg));
                SideLane* zside = nullptr;
                if (k1z) {
                    (void)k1z_lds_bytes(&hg.small_cap, false);
                    hlds = k1z_lds_bytes(&hg.box_cap, large_boxes);
                    hg.off_box = 0;
                    hg.hint = sh ? tg.hint : nullptr;
                    double r2 = 0.0;
                    for (int k = 0; k < 3; ++k) {
                        const double r = g.in_len[k] > 1 ? (double)(g.ncp[k] - 1) / (double)(g.in_len[k] - 1) : 0.0;
                        r2 += r * r;
                    }
                    zg.slack_scale = 0.15 * 1.125 * 4.0 * r2;
                    zside = side_lane(stream);
                    char* gb = zside ? (char*)geo_reserve(stream, zside, k1z_geo_bytes(g, ntiles, nb), &e) : nullptr;
                    if (!gb)
                        return e != hipSuccess ? e : hipErrorOutOfMemory;
                    zg.ctl = (int*)gb;                       // the lists' counters: 32 ints in the cleared head
                    zg.zgen = gb + 4096;
                    char* zx = gb + 4096 + 512;
                    zg.zt = (const AxTab*)zx;
                    zg.recs = (int*)(zx + k1z_zt_bytes(g));
                    zg.missed = zg.recs + (size_t)ntiles * nb * 8;
                    zg.sinfo = zg.missed + (size_t)ntiles * nb;
                    zg.list_g = zg.sinfo + (size_t)ntiles * nb;
                    zg.list_f = zg.list_g + (size_t)ntiles * nb * 8;
                    zg.recs_half = zg.list_f + (size_t)ntiles * nb * 8;
                    zg.steps = (long long*)(zx + k1z_zt_bytes(g) + (((size_t)ntiles * nb * 168 + 63) & ~(size_t)63));
                    zg.r = (const double*)((char*)zg.steps + kK1zMaxSteps * 16);
                    zg.r_bstride = (long long)(((k1z_r_bytes(g) + 63) & ~(size_t)63) / 8);
                    zg.parity = (int)(zside->parity & 1);
                    zg.disp_bstride = tg.disp_bstride;
                    zg.ncpz = (int)g.ncp[0];
                    zg.order = ORDER;
                    // strips of 4 tiles where that leaves the chip enough of them (256^3: 8192); strips of 2 below ~4096 (128^3:
                    // 1024 -> 2048 strips, level-1 launch 45.8 -> 37.0 us; 192^3: 105.8 -> 91.8; at 256^3 strips of 2 cost 6 %);
                    // single tiles only for volumes that have fewer than 1024 pairs (profiles/r06_k1_route_sweep.txt)
                    zg.strip_tiles = 4;
                    auto zstrips = [&](int st) { return (int64_t)nb * tg.tiles[1] * tg.tiles[2] * ((tg.tiles[0] + st - 1) / st); };
                    if (zstrips(4) < 4096)
                        zg.strip_tiles = 2;
                    if (zstrips(2) < 1024)
                        zg.strip_tiles = 1;
#ifdef EDHIP_EXPERIMENTS
                    if (const char* st = ed_env("EDHIP_ZSTRIP"))
                        zg.strip_tiles = atoi(st) >= 1 ? atoi(st) : zg.strip_tiles;
#endif
                    zg.nstrips = tg.tiles[1] * tg.tiles[2] * ((tg.tiles[0] + zg.strip_tiles - 1) / zg.strip_tiles);
                    zg.total_strips = zg.nstrips * nb;
                    zg.hint = hg.hint;
                    zg.hint_host = tg.hint_host;
                    zg.hint_seq = tg.hint_seq;
#ifdef EDHIP_EXPERIMENTS
                    if (const char* pp = ed_env("EDHIP_DEBUG_PTR"))
                        hg.dbgbuf = (unsigned long long*)strtoull(pp, nullptr, 16);
#endif
                } else if (k1_new) {
                    int off_small = 0;
                    (void)k1_lds_bytes(hot_cols, &hg.small_cap, &off_small, false);
                    if (large_boxes)
                        hlds = k1_lds_bytes(hot_cols, &hg.box_cap, &hg.off_box, true);
                    if (!hlds)
                        hlds = k1_lds_bytes(hot_cols, &hg.box_cap, &hg.off_box, false);
                    hg.hint = sh ? tg.hint : nullptr;
                    // margin of the sampled boxes: 15 % of the rigorous interpolation bound (see the tables kernel)
                    double r2 = 0.0;
                    for (int k = 0; k < 3; ++k) {
                        const double r = g.in_len[k] > 1 ? (double)(g.ncp[k] - 1) / (double)(g.in_len[k] - 1) : 0.0;
                        r2 += r * r;
                    }
                    tg.slack = (double*)((char*)ws + 2 * list_bytes + xt_only_bytes(g));
                    tg.slack_scale = 0.15 * 1.125 * 4.0 * r2;
                    hg.slack = tg.slack;
#ifdef EDHIP_EXPERIMENTS
                    if (const char* pp = ed_env("EDHIP_DEBUG_PTR"))
                        hg.dbgbuf = (unsigned long long*)strtoull(pp, nullptr, 16);
#endif
                } else {
                    int off_small = 0;
                    (void)hot_lds_bytes(GRAD, hot_cols, &hg.small_cap, &off_small, 0);
                    (void)hot_lds_bytes(GRAD, hot_cols, &hg.large_cap, &off_small, 1);
                    if (huge_boxes)
                        hlds = hot_lds_bytes(GRAD, hot_cols, &hg.box_cap, &hg.off_box, 2);
                    if (!hlds && large_boxes)
                        hlds = hot_lds_bytes(GRAD, hot_cols, &hg.box_cap, &hg.off_box, 1);
                    if (!hlds)
                        hlds = hot_lds_bytes(GRAD, hot_cols, &hg.box_cap, &hg.off_box, 0);
                    hg.hint = sh ? tg.hint : nullptr;
                }
                if (v.out16 && !hlds)
                    return hipErrorNotSupported;        // (nothing has been launched yet)
                // (a wide grid whose window leaves no room for a hot box: the general kernels below, same tables)
                hg.lds_grp = (int)((hlds + 15) & ~(size_t)15);
                // EDHIP_FLAG_KEEP_BOXES / USE_BOXES: the forward kernel's tile boxes -- and, orders 1-3, its
                // per-voxel coordinate records -- live in a buffer of their own (nothing else writes it) under
                // a host-side key of everything they depend on.  Layout: boxes | flags | grid copy | records.
                KeepKey* key = nullptr;
                KeepKey cur;
                memset(&cur, 0, sizeof(cur));
                const int box_mode = batch ? batch->box_mode : 0;
                const size_t ngrid = 3 * (size_t)g.ncp[0] * (size_t)g.ncp[1] * (size_t)g.ncp[2];
                const size_t nvox = (size_t)g.out_len[0] * (size_t)g.out_len[1] * (size_t)g.out_len[2];
                const size_t kb_boxes = ((size_t)ntiles * nb * 8 * sizeof(int) + 255) & ~(size_t)255;
                const size_t kb_flags = ((size_t)nb * sizeof(int) + 255) & ~(size_t)255;
                const size_t kb_stash = ((size_t)nb * ngrid * sizeof(double) + 255) & ~(size_t)255;
                // records: orders 1-3 on the 4-wave kernels (the one-wave kernels of orders 4 / 5 keep to boxes)
                // Measured (profiles/r04_records_route.txt): with the records the gradient kernel of the benchmark
                // step takes 290-305 us against hot_grad_kernel's 303-306, and K1 pays 15-20 us for writing them;
                // the route stays in the tree for the profiling build (EDHIP_RECORDS=1), the shipped library keeps
                // the boxes-only hand-over.
                bool want_rec = ORDER <= 3 && ngrid <= 65536 && (double)nvox * nb * 16.0 <= (double)((size_t)16 << 30) &&
                                ed_env("EDHIP_RECORDS") != nullptr;
#ifdef EDHIP_EXPERIMENTS
                if (ed_env("EDHIP_WAVE"))
                    want_rec = false;
#endif
                // a gradient call takes the records route whenever it can: with the forward call's records
                // when the key matches (and the tables kernel finds the grid values unchanged), with its own
                // records-only launch otherwise
                const bool rec_grad = GRAD && want_rec;
                bool rec_route = false;
                if (hlds && (box_mode || rec_grad) && !ed_env("EDHIP_NO_BOXES")) {
                    hipError_t ke = hipSuccess;
                    const bool with_rec = want_rec && (rec_grad || box_mode == 1);
                    char* kbuf = (char*)keep_reserve(stream, kb_boxes + kb_flags + kb_stash + (with_rec ? nvox * nb * 16 : 0),
                                                     &key, &ke);
                    if (kbuf) {
                        int* kb = (int*)kbuf;
                        int* kflags = (int*)(kbuf + kb_boxes);
                        double* kstash = (double*)(kbuf + kb_boxes + kb_flags);
                        float4* krec = (float4*)(kbuf + kb_boxes + kb_flags + kb_stash);
                        int n = 0;
                        cur.w[n++] = 1;                                   // valid
                        cur.w[n++] = (unsigned long long)(size_t)batch->disp_id;
                        cur.w[n++] = (unsigned long long)g.disp_dtype | ((unsigned long long)batch->raw << 8) |
                                     ((unsigned long long)ORDER << 16) | ((unsigned long long)tg.mode << 24) |
                                     ((unsigned long long)g.has_affine << 32) | ((unsigned long long)with_rec << 40);
                        cur.w[n++] = (unsigned long long)nb;
                        cur.w[n++] = (unsigned long long)batch->disp_bstride;
                        for (int k = 0; k < 3; ++k) {
                            cur.w[n++] = (unsigned long long)g.in_len[k];
                            cur.w[n++] = (unsigned long long)g.out_len[k];
                            cur.w[n++] = (unsigned long long)g.off[k];
                            cur.w[n++] = (unsigned long long)g.ncp[k];
                        }
                        for (int k = 0; k < 4; ++k)      // (RAW: the strides of the filtered copy -- a function of ncp)
                            cur.w[n++] = (unsigned long long)g.disp_stride[k];
                        if (g.has_affine)
                            for (int k = 0; k < 12; ++k)
                                memcpy(&cur.w[n++], &g.affine[k], sizeof(double));
                        if (!GRAD && box_mode == 1) {
                            hg.boxes = kb;
                            if (with_rec) {
                                hg.rec = krec;
                                hg.rec_bstride = (long long)nvox;
                                tg.keep_mode = 1;
                                tg.keep_stash = kstash;
                            }
                            *key = KeepKey();                  // not valid until this launch is in the stream
                        } else if (rec_grad) {
                            const bool match = box_mode == 2 && memcmp(key, &cur, sizeof(cur)) == 0;
                            hg.boxes = kb;
                            hg.rec = krec;
                            hg.rec_bstride = (long long)nvox;
                            hg.rec_valid = match ? kflags : nullptr;
                            if (match) {
                                tg.keep_mode = 2;
                                tg.keep_stash = kstash;
                                tg.keep_flags = kflags;
                            } else {
                                *key = KeepKey();              // this call's own records replace whatever was kept
                            }
                            rec_route = true;
                        } else if (GRAD && box_mode == 2 && ORDER >= 3 && memcmp(key, &cur, sizeof(cur)) == 0) {
                            // (orders 1 / 2: the box pass is a small part of a short tile, and the
                            // hand-over's bookkeeping costs more than it saves: 190 -> 196, 242 -> 252 us)
                            hg.boxes = kb;
                            hg.use_boxes = 1;
                            if (self_serve_boxes && ORDER <= 3)
                                hg.self_serve = 1;
                        }
                    }
                }
                if (k1z && e == hipSuccess) {
                    // the geometry kernel in place of the tables kernel: R, the z table, tile records (and boxes), work lists
                    GridPrefilter gp;
                    memset(&gp, 0, sizeof(gp));
                    const bool own = batch && batch->gridpf && nb == 1 && batch->gridpf->total <= 4096;
                    if (own) {
                        gp = *batch->gridpf;
                        gp.zero_ptr = nullptr;
                        gp.zero_bytes = 0;
                    }
                    if (!tables_done && e == hipSuccess) {
                        e = launch_k1z_geo(g, hg, zg, gp, nb, stream);
                        if (e == hipSuccess)
                            ++zside->parity;       // (the launch is in the stream: the next call uses the other counters)
                        if (own && e == hipSuccess)
                            batch->gridpf_done = true;
                        tables_done = true;
                    }
                }
                // round 6, gradient, PROFILING BUILD ONLY: the scatter on the z-walk tables (tools/experiments/
                // deform_k2z.hip; EDHIP_K2Z / EDHIP_K2Y / EDHIP_K2S pick the variant) for the geometries that serve
                // themselves.  Measured against hot_grad_kernel (profiles/r06_k2_zwalk.txt): 327 / 514 / 391 against 323 us
                // per gradient call -- not shipped; everything the shipped library launches here is tables + hot_grad_kernel
                bool k2z = false;
#ifdef EDHIP_EXPERIMENTS
                if constexpr (GRAD && ORDER <= 3) {
                    k2z = hlds && !rec_route && !wide && !v.out16 && hg.self_serve && k1z_supported(g) &&
                          ve.nsteps <= (int64_t)kK1zMaxSteps && (ed_env("EDHIP_K2Z") || ed_env("EDHIP_K2Y") || ed_env("EDHIP_K2S")) &&
                          !ed_env("EDHIP_WAVE") && e == hipSuccess;
                }
#endif
                size_t k2lds = 0;
                if (k2z) {
                    zside = side_lane(stream);
                    char* gb = zside ? (char*)geo_reserve(stream, zside, k1z_geo_bytes(g, ntiles, nb), &e) : nullptr;
                    if (!gb)
                        return e != hipSuccess ? e : hipErrorOutOfMemory;
                    zg.ctl = (int*)gb;
                    zg.zgen = gb + 4096;
                    char* zx = gb + 4096 + 512;
                    zg.zt = (const AxTab*)zx;
                    zg.steps = (long long*)(zx + k1z_zt_bytes(g) + (((size_t)ntiles * nb * 168 + 63) & ~(size_t)63));
                    zg.r = (const double*)((char*)zg.steps + kK1zMaxSteps * 16);
                    zg.r_bstride = (long long)(((k1z_r_bytes(g) + 63) & ~(size_t)63) / 8);
                    zg.disp_bstride = tg.disp_bstride;
                    zg.ncpz = (int)g.ncp[0];
                    zg.order = ORDER;
                    zg.tables_only = 1;
                    zg.strip_tiles = 4;
                    while (zg.strip_tiles > 1 && (int64_t)nb * tg.tiles[1] * ((tg.tiles[2] + 1) / 2) *
                                                         ((tg.tiles[0] + zg.strip_tiles - 1) / zg.strip_tiles) < 1024)
                        zg.strip_tiles >>= 1;
                    zg.hint = hg.hint;
                    zg.hint_host = tg.hint_host;
                    zg.hint_seq = tg.hint_seq;
                    if (batch && batch->zero_ptr && !batch->zero_done && nb == 1 && ((uintptr_t)batch->zero_ptr & 15) == 0) {
                        zg.zero_ptr = batch->zero_ptr;
                        zg.zero_bytes = batch->zero_bytes;
                    }
                    GridPrefilter gp;
                    memset(&gp, 0, sizeof(gp));
                    const bool own = batch && batch->gridpf && nb == 1 && batch->gridpf->total <= 4096;
                    if (own) {
                        gp = *batch->gridpf;
                        gp.zero_ptr = nullptr;
                        gp.zero_bytes = 0;
                    }
#ifdef EDHIP_EXPERIMENTS
                    k2lds = k2z_lds_bytes(&hg.box_cap);
#endif
                    hg.small_cap = hg.box_cap;
                    e = launch_k1z_geo(g, hg, zg, gp, nb, stream);
                    if (e == hipSuccess) {
                        if (zg.zero_ptr)
                            batch->zero_done = true;
                        if (own)
                            batch->gridpf_done = true;
                    }
                    tables_done = true;
                }
                if (!tables_done)
                    launch_tables();
#ifdef EDHIP_EXPERIMENTS
                if (k2z && e == hipSuccess) {
                    profile_mark(false, stream);
                    e = launch_k2z(hg, zg, ORDER, k2lds, stream);
                    hot_done = true;
                    served_all = true;
                }
#endif
                if (rec_route && e == hipSuccess) {
                    // gradient from records: (1) K1 in records-only form -- its workgroups leave at once for the
                    // samples whose records the forward call made from these very grid values -- (2) the
                    // gradient kernel that reads records and boxes
                    HotGeom rg = hg;
                    rg.self_serve = 0;
                    int fcap = 0, foff = 0;
                    const size_t flds = hot_lds_bytes(false, tg.ncpx, &fcap, &foff, false);
                    rg.box_cap = fcap;
                    rg.off_box = foff;
                    rg.small_cap = fcap;
                    rg.hint = nullptr;
                    rg.rec_only = 1;
                    hipError_t he = flds ? launch_hot_records(rg, ORDER, nblk, flds, stream) : hipErrorNotSupported;
                    if (he == hipSuccess) {
                        HotGeom gg = hg;
                        gg.self_serve = 0;
                        gg.rec_valid = nullptr;
                        int gcap = 0, scap = 0;
                        (void)hot_grad2_lds_bytes(&scap, false);
                        const size_t glds = hot_grad2_lds_bytes(&gcap, large_boxes);
                        gg.box_cap = gcap;
                        gg.small_cap = scap;
#ifdef EDHIP_EXPERIMENTS
                        if (const char* pp = ed_env("EDHIP_DEBUG_PTR"))
                            gg.dbgbuf = (unsigned long long*)strtoull(pp, nullptr, 16);
#endif
                        profile_mark(false, stream);
                        he = launch_hot_grad2(gg, ORDER, nblk, glds, stream);
                        if (he == hipSuccess)
                            hot_done = true;
                    }
                    if (he != hipSuccess && he != hipErrorNotSupported)
                        e = he;
                    if (!hot_done) {          // (cannot happen for orders 1-3; the old route needs its own state)
                        hg.boxes = nullptr;
                        hg.rec = nullptr;
                    }
                }
                if (!hot_done)
                    profile_mark(false, stream);
                // Orders 4 / 5 run on the wave-per-tile kernels (deform_wave.hip: one wavefront per tile,
                // one copy of the box, tiles that do not fit taken as two x-halves): the 4-wave kernels
                // give up on a tile as soon as its 5- / 6-tap windows need the wide row pitch, and at
                // these orders that is a large share of the tiles (256^3, sigma 5, whole call: order 4
                // forward 419 -> 339 us, order 5 gradient 944 -> 886 us; profiles/r03_wave_vs_4wave.txt).
                // Orders 1-3 stay on the 4-wave kernels, which are 10-30 % faster there.
                bool wave_done = false;
                int wave_mode = ORDER >= 4 ? 3 : 0;
#ifdef EDHIP_EXPERIMENTS
                if (const char* wm = ed_env("EDHIP_WAVE"))        // 1 forward, 2 gradient, 3 both, 0 neither
                    wave_mode = atoi(wm);
#endif
                if ((hlds || wide_wave) && !hot_done && (wave_mode & (GRAD ? 2 : 1))) {
                    HotGeom wg = hg;
                    wg.self_serve = 0;
                    wg.strip_tiles = 4;
#ifdef EDHIP_EXPERIMENTS
                    if (const char* st = ed_env("EDHIP_WAVE_STRIP"))
                        wg.strip_tiles = atoi(st);
#endif
                    while (wg.strip_tiles > 1 &&
                           (int64_t)nb * tg.tiles[0] * tg.tiles[1] * ((tg.tiles[2] + wg.strip_tiles - 1) / wg.strip_tiles) < 8192)
                        wg.strip_tiles >>= 1;
                    wg.strips_x = (tg.tiles[2] + wg.strip_tiles - 1) / wg.strip_tiles;
                    const int64_t wstrips = (int64_t)tg.tiles[0] * tg.tiles[1] * wg.strips_x;
                    if (wstrips * nb <= 0x3fffffffLL) {
                        wg.nstrips = (int)wstrips;
                        wg.total_strips = (int)(wstrips * nb);
                        int occ = 3;
#ifdef EDHIP_EXPERIMENTS
                        if (const char* oc = ed_env("EDHIP_WAVE_OCC"))
                            occ = atoi(oc);
#endif
                        size_t wlds = wave_lds_bytes(GRAD, occ, &wg.box_cap);
                        // spill feedback: larger boxes (fewer waves per CU) once the geometry's calls spill.
                        // 256^3 sigma 10, whole call: order 4 forward 453 -> 420 us, gradient 694 -> 662;
                        // order 5 forward 1019 -> 657, gradient 1238 -> 1055 (sigma 5: +20 .. +40 us)
                        wg.small_cap = wg.box_cap;
                        wg.hint = sh ? tg.hint : nullptr;
                        if (large_boxes) {
                            wlds = (ORDER == 5 && !GRAD) ? 20480 : 16384;
                            wg.box_cap = (int)((wlds - 416) / 4);
                        }
#ifdef EDHIP_EXPERIMENTS
                        if (const char* kb = ed_env("EDHIP_WAVE_LDS")) {
                            wlds = (size_t)atoi(kb);
                            wg.box_cap = (int)((wlds - 416) / 4);
                        }
                        if (const char* pp = ed_env("EDHIP_DEBUG_PTR"))
                            wg.dbgbuf = (unsigned long long*)strtoull(pp, nullptr, 16);
#endif
                        const unsigned wblk = (unsigned)(((wstrips * nb + 7) / 8) * 8);
                        const hipError_t he = launch_wave_level1(wg, ORDER, GRAD, wblk, wlds, occ, stream);
                        if (he == hipSuccess) {
                            hot_done = wave_done = true;
                            if (!GRAD && hg.boxes && key)
                                *key = cur;
                        } else if (he != hipErrorNotSupported || wide_wave)
                            e = he;
                    }
                }
                if (wide_wave && !wave_done && e == hipSuccess)
                    e = hipErrorNotSupported;        // (no other level-1 kernel can take a grid this wide)
                if (hlds && !wave_done && !hot_done && e == hipSuccess) {
                    const hipError_t he = k1z ? launch_k1z(hg, zg, ORDER, hlds, stream, zside)
                                          : (k1_new ? launch_k1_level1(hg, ORDER, nblk, hlds, stream)
                                                    : launch_hot_level1(hg, ORDER, GRAD, nblk, hlds, stream));
                    if (he == hipSuccess) {
                        hot_done = true;
                        served_all = hg.self_serve != 0 || k1z;
                        if (!GRAD && hg.boxes && key)
                            *key = cur;
                    } else if (he != hipErrorNotSupported || v.out16 || wide)
                        e = he;
                }

            }
        }
        if (!tables_done) {
            launch_tables();
            profile_mark(false, stream);
        }
        if (hot_done || e != hipSuccess)
            ;
        else if (GRAD)
            hipLaunchKernelGGL((deform_tile3_grad_kernel<T, (ORDER < 1 ? 2 : ORDER), 16>), dim3(nblk),
                               dim3(kBlock), lds, stream, g, ve, tg);
#ifdef EDHIP_EXPERIMENTS
        else if (kBenchKernel && tg.dbg) {
            if constexpr (kBenchKernel) {
            // profiling builds of the benchmark kernel (EDHIP_TILE_DBG), never used otherwise
            switch (tg.dbg) {
            case 2: hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR, 2>), dim3(nblk), dim3(kBlock), lds, stream, g, ve, tg); break;
            case 4: hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR, 4>), dim3(nblk), dim3(kBlock), lds, stream, g, ve, tg); break;
            case 6: hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR, 6>), dim3(nblk), dim3(kBlock), lds, stream, g, ve, tg); break;
            case 16: hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR, 16>), dim3(nblk), dim3(kBlock), lds, stream, g, ve, tg); break;
            case 64: hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR, 64>), dim3(nblk), dim3(kBlock), lds, stream, g, ve, tg); break;
            case 38: hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR, 38>), dim3(nblk), dim3(kBlock), lds, stream, g, ve, tg); break;
            default: hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR, 0>), dim3(nblk), dim3(kBlock), lds, stream, g, ve, tg); break;
            }
            }
        }
#endif
        else
            hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR>), dim3(nblk), dim3(kBlock),
                               lds, stream, g, ve, tg);
        e = hipGetLastError();
    }
    profile_mark(true, stream);
    // ---- level 2: the tiles level 1 could not hold, one 8^3 tile per work item, a 48 KiB box
    //      (single copy, 24- / 56-wide rows), grid-stride over the worklist --------------------------
    TileGeom t2 = tg;
    t2.worklist = list_a;
    t2.spill = list_b;
    // (float32 up to order 3: 32 KiB boxes hold nearly every tile level 1 gives up on, and three
    // workgroups per CU instead of two serve them a third faster -- 256^3 sigma 10: forward call
    // 404 -> 368 us, gradient call 532 -> 502 us; smaller boxes send tiles to the direct level, which is
    // far slower for gradients: profiles/r03_bench_misc.txt)
    size_t box2 = (sizeof(T) == 4 && ORDER <= 3) ? 32 * 1024 : 48 * 1024;
    unsigned wgs2 = (sizeof(T) == 4 && ORDER <= 3) ? 768 : 512;
    if (const char* kb = ed_env("EDHIP_L2_BOX_KB"))        // experiment: smaller boxes, more workgroups per CU
        box2 = (size_t)atoi(kb) * 1024;
    if (const char* w = ed_env("EDHIP_L2_WGS"))
        wgs2 = (unsigned)atoi(w);
    if (t2.off_ov + box2 > 64 * 1024)
        box2 = (64 * 1024 - t2.off_ov) & ~(size_t)63;
    t2.box_cap = (int)((box2 - (GRAD ? 64 : 0)) / sizeof(T));      // gradient: cells of sizeof(T) + wave sums
    const size_t lds2 = t2.off_ov + box2;
    const unsigned n2 = (unsigned)(ntiles * nb < wgs2 ? ntiles * nb : wgs2);
    const bool skip_l2 = ed_env("EDHIP_SKIP_L2") != nullptr || wide_wave;      // (debugging aid; wide grids: see above)
    if (served_all)
        return e;               // level 1 has served every tile
    if (e == hipSuccess && !skip_l2) {
        if (GRAD)
            hipLaunchKernelGGL((deform_tile3_grad_kernel<T, (ORDER < 1 ? 2 : ORDER), 8>), dim3(n2),
                               dim3(kBlock), lds2, stream, g, ve, t2);
        else
            hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, false>), dim3(n2), dim3(kBlock), lds2,
                               stream, g, ve, t2);
        e = hipGetLastError();
    }
    // ---- level 3: whatever is left goes straight through global memory ----------------------------
    if (e == hipSuccess) {
        TileGeom t3 = tg;
        t3.spill = skip_l2 ? list_a : list_b;
        const unsigned nsp = (unsigned)(ntiles * nb < 512 ? ntiles * nb : 512);
        hipLaunchKernelGGL((deform_tile3_direct_kernel<T, ORDER, GRAD, true>), dim3(nsp), dim3(kBlock),
                           0, stream, g, ve, t3);
        e = hipGetLastError();
    }
    if (e == hipSuccess && ed_env("EDHIP_PRINT_SPILL")) {      // debugging aid: tiles per level
        int c1 = 0, c2 = 0;
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(&c1, list_a, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&c2, list_b, 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "edhip: %s tiles %lld, level-2 %d, level-3 %d\n", GRAD ? "K2" : "K1",
                (long long)(ntiles * nb), c1, c2);
    }
    return e;
    }   // floating point T
}

}  // namespace

void tile_profile_enable(int enable) { g_profile.store(enable ? 1 : 0); }

double tile_profile_last_us()
{
    if (!t_ev_pending || !t_ev0)
        return -1.0;
    float ms = 0.f;
    if (hipEventSynchronize(t_ev1) != hipSuccess || hipEventElapsedTime(&ms, t_ev0, t_ev1) != hipSuccess)
        return -1.0;
    t_ev_pending = false;
    return (double)ms * 1e3;
}

size_t deform_tile_workspace_bytes(const GridGeom& g, int nbatch, bool f64)
{
    if (g.naxis != 3)
        return kWorkspaceGridBytes;
    int64_t ntiles = 1;
    for (int k = 0; k < 3; ++k)
        ntiles *= (g.out_len[k] + kT - 1) / kT;
    const size_t list_bytes = (sizeof(int) * ((size_t)ntiles * nbatch + 1) + 63) & ~(size_t)63;
    const size_t xt_bytes = xt_block_bytes(g, nbatch);
    const size_t q = q_global_bytes(g, f64 && nbatch == 1);       // (float64 volumes may take the per-strip layout)
    if (q > ((size_t)512 << 20))
        return kWorkspaceGridBytes;
    return kWorkspaceGridBytes + 2 * list_bytes + xt_bytes + ((q * (size_t)nbatch + 63) & ~(size_t)63) +
           label_list_bytes(g);
}

bool deform_tile_supported(const GridGeom& g, const IOView& v, int gradient)
{
    if (g.naxis != 3)
        return false;
    if (!deform_fast_supported(g, v, gradient))
        return false;
    const int64_t esz = v.in_dtype == EDHIP_F32 ? 4 : 8;
    int64_t in_span = 0, out_span = 0;
    for (int k = 0; k < 3; ++k) {
        if (g.in_len[k] >= 0x3fffffff || g.out_len[k] >= 0x3fffffff || g.ncp[k] > 1024)
            return false;
        const int64_t si = v.in_stride[k] / esz, so = v.out_stride[k] / esz;
        in_span += (si < 0 ? -si : si) * (g.in_len[k] - 1);
        out_span += (so < 0 ? -so : so) * (g.out_len[k] - 1);
    }
    if (in_span >= 0x7fffffffLL || out_span >= 0x7fffffffLL)   // 32-bit element offsets inside a volume
        return false;
    // head + Q + box must stay within a 64 KiB block; the per-call Q table within 512 MiB; the
    // tables kernel keeps 3 * ncp_y * ncp_x doubles in LDS.  Wider control grids: float32 volumes of orders 1-3 with
    // unit stride along x go to the level-1 kernels on per-strip Q tables (launch_tile), everything else to the
    // row kernel of deform_fast.hip
    if (wide_grid(g)) {
        if (v.order < 1)
            return false;
        // float32 orders 4 / 5 with unit stride along x: the one-wave kernels, plain tables, any width; everything else
        // on per-strip tables, whose window has to fit
        const bool wave = v.in_dtype == EDHIP_F32 && v.order >= 4 && v.in_stride[2] == 4 && v.out_stride[2] == 4;
        if (!wave &&
            (wide_window(g) > kWideMaxWin || (g.out_len[2] + kWideStripTiles * kT - 1) / (kWideStripTiles * kT) > 256))
            return false;
    }
    if (q_global_bytes(g) > ((size_t)512 << 20) || 24 * (size_t)g.ncp[1] * (size_t)g.ncp[2] > 48 * 1024)
        return false;
    return true;
}

static int label_elem_size(int dt)
{
    switch (dt) {
    case EDHIP_BOOL: case EDHIP_U8: case EDHIP_I8: return 1;
    case EDHIP_U16: case EDHIP_I16: case EDHIP_F16: case EDHIP_BF16: return 2;
    case EDHIP_U32: case EDHIP_I32: case EDHIP_F32: return 4;
    default: return 8;
    }
}

bool deform_label_supported(const GridGeom& g, const IOView& v, int gradient)
{
    if (g.naxis != 3 || gradient || v.order != 0 || v.in_dtype != v.out_dtype)
        return false;
    const int64_t esz = label_elem_size(v.in_dtype);
    if (((uintptr_t)v.in % esz) || ((uintptr_t)v.out % esz))
        return false;
    int64_t in_span = 0, out_span = 0;
    for (int k = 0; k < 3; ++k) {
        if (g.in_len[k] >= 0x3fffffff || g.out_len[k] >= 0x3fffffff || g.ncp[k] > 1024)
            return false;
        if (v.in_stride[k] % esz || v.out_stride[k] % esz)
            return false;
        const int64_t si = v.in_stride[k] / esz, so = v.out_stride[k] / esz;
        in_span += (si < 0 ? -si : si) * (g.in_len[k] - 1);
        out_span += (so < 0 ? -so : so) * (g.out_len[k] - 1);
    }
    for (int l = 0; l < v.nstep; ++l)
        if (v.in_step_stride[l] % esz || v.out_step_stride[l] % esz)
            return false;
    if (in_span >= 0x7fffffffLL || out_span >= 0x7fffffffLL)
        return false;
    if (q_global_bytes(g) > ((size_t)512 << 20) || 24 * (size_t)g.ncp[1] * (size_t)g.ncp[2] > 48 * 1024)
        return false;
    // wide control grids (per-strip Q tables) are a float route: label maps and the integer fast path stay on whole-grid
    // tables, beyond them the exact kernels take the volume.  (This test was missing: launch_tile declined such a call
    // AFTER the route had been chosen and edhip_deform failed with "operation not supported" -- found by fuzz_api.py.)
    if (wide_grid(g))
        return false;
    return true;
}

// integer fast path: 8- / 16-bit volumes, orders 1-5, forward, unit stride along the last deformed axis
bool deform_int_supported(const GridGeom& g, const IOView& v, int gradient)
{
    if (g.naxis != 3 || gradient || v.order < 1 || v.order > 5 || v.in_dtype != v.out_dtype)
        return false;
    if (v.in_dtype != EDHIP_U8 && v.in_dtype != EDHIP_I8 && v.in_dtype != EDHIP_U16 && v.in_dtype != EDHIP_I16 &&
        v.in_dtype != EDHIP_U32 && v.in_dtype != EDHIP_I32)
        return false;
    IOView v0 = v;
    v0.order = 0;
    if (!deform_label_supported(g, v0, 0))
        return false;
    const int64_t esz = label_elem_size(v.in_dtype);
    if (v.in_stride[2] != esz || v.out_stride[2] != esz)
        return false;
    for (int k = 0; k < 3; ++k)
        if (g.in_len[k] < 2)
            return false;
    return true;
}

// (batch: nullptr, or a single-volume DeformBatch that carries the raw control grid for the tables kernel to filter)
hipError_t launch_deform_int(const GridGeom& g, const IOView& v, hipStream_t stream, const DeformBatch* batch)
{
    if (!deform_int_supported(g, v, 0))
        return hipErrorNotSupported;
    switch (label_elem_size(v.in_dtype)) {
    case 1: return launch_tile<uint8_t, 0, false, false>(g, v, stream, batch);
    case 2: return launch_tile<uint16_t, 0, false, false>(g, v, stream, batch);
    default: return launch_tile<uint32_t, 0, false, false>(g, v, stream, batch);
    }
}

hipError_t launch_deform_label(const GridGeom& g, const IOView& v, hipStream_t stream, const DeformBatch* batch)
{
    if (!deform_label_supported(g, v, 0))
        return hipErrorNotSupported;
    switch (label_elem_size(v.in_dtype)) {
    case 1: return launch_tile<uint8_t, 0, false, false>(g, v, stream, batch);
    case 2: return launch_tile<uint16_t, 0, false, false>(g, v, stream, batch);
    case 4: return launch_tile<uint32_t, 0, false, false>(g, v, stream, batch);
    default: return launch_tile<uint64_t, 0, false, false>(g, v, stream, batch);
    }
}

hipError_t launch_deform_tile(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream,
                              const DeformBatch* batch)
{
    if (!deform_tile_supported(g, v, gradient))
        return hipErrorNotSupported;
    if (batch && (double)q_global_bytes(g) * batch->nbatch > (double)((size_t)4 << 30))
        return hipErrorNotSupported;          // per-sample Q tables: keep the scratch bounded
    const bool f32 = v.in_dtype == EDHIP_F32;
    if (v.order < 2) {
        // direct kernel (forward gathers / float atomics)
        if (gradient) {
            if (f32)
                return v.order == 0 ? launch_tile<float, 0, false, true>(g, v, stream, batch)
                                    : launch_tile<float, 1, false, true>(g, v, stream, batch);
            return v.order == 0 ? launch_tile<double, 0, false, true>(g, v, stream, batch)
                                : launch_tile<double, 1, false, true>(g, v, stream, batch);
        }
        if (f32)
            return v.order == 0 ? launch_tile<float, 0, false, false>(g, v, stream, batch)
                                : launch_tile<float, 1, true, false>(g, v, stream, batch);
        return v.order == 0 ? launch_tile<double, 0, false, false>(g, v, stream, batch)
                            : launch_tile<double, 1, false, false>(g, v, stream, batch);
    }
    if (gradient) {
        if (f32) {
            switch (v.order) {
            case 2: return launch_tile<float, 2, false, true>(g, v, stream, batch);
            case 3: return launch_tile<float, 3, false, true>(g, v, stream, batch);
            case 4: return launch_tile<float, 4, false, true>(g, v, stream, batch);
            case 5: return launch_tile<float, 5, false, true>(g, v, stream, batch);
            default: return hipErrorNotSupported;
            }
        }
        switch (v.order) {
        case 2: return launch_tile<double, 2, false, true>(g, v, stream, batch);
        case 3: return launch_tile<double, 3, false, true>(g, v, stream, batch);
        case 4: return launch_tile<double, 4, false, true>(g, v, stream, batch);
        case 5: return launch_tile<double, 5, false, true>(g, v, stream, batch);
        default: return hipErrorNotSupported;
        }
    }
    if (f32) {
        switch (v.order) {
        case 2: return launch_tile<float, 2, true, false>(g, v, stream, batch);
        case 3: return launch_tile<float, 3, true, false>(g, v, stream, batch);
        case 4: return launch_tile<float, 4, true, false>(g, v, stream, batch);
        case 5: return launch_tile<float, 5, true, false>(g, v, stream, batch);
        default: return hipErrorNotSupported;
        }
    }
    switch (v.order) {
    case 2: return launch_tile<double, 2, false, false>(g, v, stream, batch);
    case 3: return launch_tile<double, 3, false, false>(g, v, stream, batch);
    case 4: return launch_tile<double, 4, false, false>(g, v, stream, batch);
    case 5: return launch_tile<double, 5, false, false>(g, v, stream, batch);
    default: return hipErrorNotSupported;
    }
}

}  // namespace ed
