// deform_k1z.hip -- K1 of round 6: the forward gather of the benchmark case (float32 volumes, 3 deformed axes, unit
// stride along x on both sides, spline orders 1-3; deform.c:649-924), rebuilt around what the round-5 ablation
// measured (profiles/r05_ablate_k1.txt, r05_k1_phases.txt): with gather, staging and stores switched off the round-5
// kernel still took 107 of its 180 us -- strip prologue (10 KB of Q rows + the x table copied into LDS per strip),
// sampled boxes and tile records made inside the kernel, table reads and address arithmetic per voxel.
//
// What is different:
//
//   * A workgroup WALKS ALONG z.  The displacement spline is contracted over y and x once per call (the geometry
//     kernel below): R[o_y][o_x][k_z][c].  A lane owns one (y, x) column of output voxels for its whole strip, so its
//     four control planes x three components are LANE-CONSTANT (24 VGPRs, reloaded from the L2 when the walk crosses a
//     control interval) and the z weights are WAVE-UNIFORM (a wave works on one z slice at a time): scalar loads,
//     SGPR operands of the 12 fp64 FMAs.  A voxel's displacement costs no LDS access, no table lookup and no
//     address arithmetic (round 5: 8 LDS reads per voxel + the strip's Q rows staged in LDS).
//   * NO STRIP PROLOGUE.  Nothing but the uniform parameters (416 bytes) is staged per strip; the LDS a workgroup
//     owns is all box: 5048 floats per copy instead of 3544 (fewer tiles that do not fit at sigma >= 10).
//   * TILE RECORDS COME FROM THE GEOMETRY KERNEL (k1z_geo_kernel, one small launch per call, in place of the tables
//     kernel): it contracts R for an 8 x 8 patch of columns in LDS, samples the coordinate at a 4 x 4 x 4 lattice of
//     every tile of the patch's z column, and writes the 64-byte record the tile loop needs -- box, pitch, flags,
//     base offset -- which the tile loop reads with ONE scalar load (round 5: one wave per tile sampled in the strip
//     prologue, lane 63 derived the record into LDS, every use went through v_readfirstlane).
//
// Unchanged, on purpose: 8^3 tiles, a lane owns a (y, x) column and two z slices of a tile, two copies of the source
// box in LDS one element apart (every x-run of taps is aligned 8-byte reads), plane-wise LDS-DMA staging, the software
// pipeline (coordinates of tile t + 1 under the copies of tile t), the per-voxel containment check of the sampled
// box with a fix-up pass behind the loop (correctness never rests on the margin), the separable gather x, y, z from
// zero, streaming stores.
//
// Bits: the displacement is summed in a different order than in the kernels that contract over z and y first
// (Q tables: deform_tile.hip, deform_k1.hip, deform_hot.hip) -- coordinates differ by ~1e-15 between the two families.
// Everything a call of this route computes (tile loop and fix-up) uses R, so a voxel gets the same bits whichever
// part of THIS kernel serves it and the crop identity full[crop] == cropped (README.md:113) holds bit for bit
// between calls of this route.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "ed_device.h"
#include "ed_gridfilter.h"
#include "ed_params.h"
#include "ed_tile.h"
#include "ed_workspace.h"
#include "ed_zwalk.h"

namespace ed {
namespace tile {

#ifdef EDHIP_K1_STATS
// (-DEDHIP_K1_STATS, tools/k1_stats.py) [0] waves that ran the fix-up for a window outside a
// sampled box, [1] voxels they redid, [2] voxels of unfit tiles, [3] class-A tiles, [4] general tiles, [5] unfit tiles,
// [6] strips the fast kernel flagged, [7] workgroups of the rest kernel with work
__device__ unsigned long long g_k1z_stats[8];
#define ZSTAT(K, N) atomicAdd(&g_k1z_stats[K], (unsigned long long)(N))
#else
#define ZSTAT(K, N) do { } while (0)
#endif

namespace {


// tile record flags (ZRec::flags)
enum : int {
    kZAny = 1,          // some voxel of the tile is gathered
    kZStaged = 2,       // ... and the box fits: staged, gathered from LDS
    kZFast = 4,         // full tile, coordinates inside the array: no boundary tests
    kZDma = 8,          // the box rows (and the shifted copy's extra element) lie inside the array along x: LDS-DMA
    kZXin = 16,         // the box lies inside the array along x
    kZZYin = 32,        // ... and along z and y: no mirror map of plane / row indices while staging
    kZUnfit = 64,       // the box does not fit: k1z_fix gathers the tile from global memory
    kZGen = 128,        // general tile (array faces, partial tiles): boundary map, constant / valid flags
};
// what the tile loops need to know about a tile: 8 dwords, one s_load_dwordx8
//   [0..2] box origin, in window-start (tap index) space      [3] flags | pitch << 8 | (beyond the standard box) << 16
//   [4] extents ez | ey << 10 | ex << 20                       [5] element offset of the box origin in the volume (class A)
struct ZRecU {
    int b0z, b0y, b0x, flags, pitch, ez, ey, ex, goff, plane;
};
typedef int zv8i __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) zv8i* crec_p;
__device__ __forceinline__ ZRecU zrec_unpack(const zv8i& v)
{
    ZRecU r;
    r.b0z = v[0];
    r.b0y = v[1];
    r.b0x = v[2];
    r.flags = v[3];
    r.pitch = (v[3] >> 8) & 255;
    r.ez = v[4] & 1023;
    r.ey = (v[4] >> 10) & 1023;
    r.ex = (int)((unsigned)v[4] >> 20);
    r.goff = v[5];
    r.plane = r.ey * r.pitch;
    return r;
}
__device__ __forceinline__ ZRecU zrec_load(crec_p p)
{
    const zv8i v = *p;
    return zrec_unpack(v);
}

// work-list counters (ZGeom::ctl): [parity][list: 0 = G, 1 = F][XCD]; entry k of XCD x's list sits at slot 8 k + x
__host__ __device__ __forceinline__ int zctl(int parity, int list, int xcd) { return (parity * 2 + list) * 8 + xcd; }

#define ED_RED6(CTRL)                                      \
    "v_min_i32_dpp %0, %0, %0 " CTRL "\n\t"                \
    "v_min_i32_dpp %1, %1, %1 " CTRL "\n\t"                \
    "v_min_i32_dpp %2, %2, %2 " CTRL "\n\t"                \
    "v_max_i32_dpp %3, %3, %3 " CTRL "\n\t"                \
    "v_max_i32_dpp %4, %4, %4 " CTRL "\n\t"                \
    "v_max_i32_dpp %5, %5, %5 " CTRL "\n\t"
// min of lo[3] / max of hi[3] over each ROW of 16 lanes (all 64 active): afterwards every lane holds its row's result
__device__ __forceinline__ void zrow_box(int (&lo)[3], int (&hi)[3])
{
    asm volatile("s_nop 1\n\t"
                 ED_RED6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 ED_RED6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_half_mirror row_mask:0xf bank_mask:0xf")
                 ED_RED6("row_mirror row_mask:0xf bank_mask:0xf")
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
}
#undef ED_RED6
__device__ __forceinline__ int zuni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void zlds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void zdma_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void zglds16(const float* g, float* lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <bool OUT16>
__device__ __forceinline__ void zstore_out(float* img, long long off, float val, int io16)
{
    if constexpr (OUT16)
        reinterpret_cast<unsigned short*>(img)[off] = (unsigned short)narrow16(val, io16);
    else
        __builtin_nontemporal_store(val, img + off);
}

// one entry of the z table / one control-axis entry (deform.c:639-647,655-690): cubic weights of the control
// coordinate + mirror-mapped control indices
__device__ __forceinline__ void zaxis_entry(const GridGeom& g, int a, int oi, double* w, int* idx)
{
    const double cp = control_coordinate(g.ncp[a], (int64_t)oi + g.off[a], g.in_len[a]);
    const int64_t start = window_start(cp, 3);
    const bool edge = start < 0 || start + 3 >= g.ncp[a];
    spline_weights(cp, 3, w);
#pragma unroll
    for (int l = 0; l < 4; ++l)
        idx[l] = edge ? mirror_i32((int)start + l, (int)g.ncp[a]) : (int)start + l;
}

constexpr int kGeoBlock = 256;            // (4 waves: the records of a column's tiles are a latency chain per tile)
constexpr int kGeoWaves = kGeoBlock / 64;
// ================================================================================================
// geometry kernel: R, the z table, the tile records (and the boxes for the gradient call)
// ================================================================================================
// One workgroup per 8 x 8 patch of (y, x) output columns and sample.  LDS: the control grid as doubles
// [3][ncp_z][ncp_y][ncp_x] | R of the patch [64 columns][ncp_z][3] | axis entries of the patch's 8 rows and 8 columns.
__global__ __launch_bounds__(kGeoBlock) void k1z_geo_kernel(const GridGeom g, const HotGeom hg, const ZGeom zg,
                                                          const GridPrefilter gp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = zuni(tid >> 6);
    const int sample = blockIdx.y;
    const int ncpz = (int)g.ncp[0], ncpy = (int)g.ncp[1], ncpx = (int)g.ncp[2];
    const int nyx = ncpy * ncpx;
    const int ngrid = 3 * ncpz * nyx;
    double* sG = reinterpret_cast<double*>(smem);
    double* sR = sG + ngrid;                                  // [64][ncpz][3]
    AxTab* sAx = reinterpret_cast<AxTab*>(sR + 64 * ncpz * 3);   // [0..8): rows (y), [8..16): columns (x)
    double* sMax = reinterpret_cast<double*>(sAx + 16);       // [3][waves] partial maxima
    AxTab* sZ = reinterpret_cast<AxTab*>(sMax + 3 * kGeoWaves);
    int* sCls = reinterpret_cast<int*>(sZ + 4 * hg.tiles[0]);        // [tiles_z]: class of every tile of the column
    int* sBox = sCls + hg.tiles[0];                                  // [tiles_z][raw | mapped][4 sampled slices][8]: sample ranges
    int* sCnt = sBox + 64 * hg.tiles[0];                             // [4]          // [tiles_z][4]: z entries of the sampled slices
    if ((int)blockIdx.x >= hg.tiles[1] * hg.tiles[2]) {
        // spare workgroups: clear the gradient accumulators (EDHIP_FLAG_ZERO_GRADIENT) while the others compute the tables
        const long long nfill = (long long)(gridDim.x - hg.tiles[1] * hg.tiles[2]) * kGeoBlock;
        const long long me = (long long)(blockIdx.x - hg.tiles[1] * hg.tiles[2]) * kGeoBlock + tid;
        const long long n16 = zg.zero_bytes >> 4;
        int4* p16 = reinterpret_cast<int4*>(zg.zero_ptr);
        for (long long i = me; i < n16; i += nfill)
            p16[i] = make_int4(0, 0, 0, 0);
        if (me < (zg.zero_bytes & 15))
            zg.zero_ptr[(n16 << 4) + me] = 0;
        return;
    }
    const int ty = blockIdx.x / hg.tiles[2], tx = blockIdx.x - ty * hg.tiles[2];
    if (ED_DBG(hg.dbg, 1 << 23))
        return;
#ifdef EDHIP_EXPERIMENTS
    // profiling build, EDHIP_DEBUG_PTR: clock of thread 0 at the phase boundaries (tools/geo_phases.py)
#define ZTICK(K) do { if (hg.dbgbuf && tid == 0) hg.dbgbuf[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 16 + (K)] = __builtin_readcyclecounter(); } while (0)
#else
#define ZTICK(K) do { } while (0)
#endif
    ZTICK(0);

    // ---- the control grid -> LDS (a RAW grid is filtered here: ed_gridfilter.h, the bits of grid_prefilter_kernel) ----
    if (gp.total > 0 && !ED_DBG(hg.dbg, 1 << 28)) {
        grid_prefilter_in_lds<kGeoBlock>(gp, sG, tid);
        if (blockIdx.x == 0 && sample == 0)
            for (int e = tid; e < gp.total; e += kGeoBlock)
                store_cast(gp.out + (int64_t)e * gp.elem_size, gp.dtype, sG[e]);
    } else {
        const char* disp = g.disp + (int64_t)sample * zg.disp_bstride;
        for (int e = tid; e < ngrid; e += kGeoBlock) {
            const int h = e / (ncpz * nyx), r = e - h * (ncpz * nyx);
            const int j0 = r / nyx, j = r - j0 * nyx;
            const int j1 = j / ncpx, j2 = j - j1 * ncpx;
            sG[e] = load_as_double(disp + g.disp_stride[0] * h + g.disp_stride[1] * j0 + g.disp_stride[2] * j1 +
                                       g.disp_stride[3] * j2, g.disp_dtype);
        }
        __syncthreads();
    }
    ZTICK(1);
    if (blockIdx.x == 0 && sample == 0 && tid == 0 && zg.hint) {
        // spill feedback (ed_workspace.h): the previous call's count goes to the host, this call's starts at zero
        if (zg.hint_host)
            __hip_atomic_store(zg.hint_host, ((unsigned long long)(unsigned)zg.hint[1] << 32) | (unsigned)zg.hint[0],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        zg.hint[0] = 0;
        zg.hint[1] = (int)zg.hint_seq;
    }
    // ---- margin of the sampled boxes: a fraction of the rigorous bound of multilinear interpolation between
    //      samples <= 3 voxels apart (see deform_k1.hip / DESIGN.md), from the largest control coefficient ----
    for (int h = 0; h < 3 && !ED_DBG(hg.dbg, 1 << 29); ++h) {
        double m = 0.0;
        for (int e = tid; e < ncpz * nyx; e += kGeoBlock)
            m = fmax(m, fabs(sG[h * ncpz * nyx + e]));
        for (int sh = 32; sh >= 1; sh >>= 1)
            m = fmax(m, __shfl_xor(m, sh));
        if (lane == 0)
            sMax[h * kGeoWaves + wave] = m;
    }
    ZTICK(2);
    // ---- axis entries of the patch's rows and columns; the z table (workgroup 0) -------------------
    if (tid < 16 && !ED_DBG(hg.dbg, 1 << 30)) {
        const int a = tid < 8 ? 1 : 2;
        const int oi = min((tid < 8 ? ty : tx) * kT + (tid & 7), hg.out_len[a] - 1);
        AxTab t;
        zaxis_entry(g, a, oi, t.w, t.idx);
        sAx[tid] = t;
    }
    for (int e = tid; e < 4 * hg.tiles[0] && !zg.tables_only && !ED_DBG(hg.dbg, 1 << 26); e += kGeoBlock) {
        const int tz = e >> 2, nz = min(kT, hg.out_len[0] - tz * kT);
        AxTab t;
        zaxis_entry(g, 0, tz * kT + ((e & 3) * (nz - 1)) / 3, t.w, t.idx);
        sZ[e] = t;
    }
    if (blockIdx.x == 0 && sample == 0 && hg.nstep) {
        // element offsets of every index of the step axes (deform.c:405-436,828-838), for the fast kernel
        for (long long ss = tid; ss < hg.nsteps; ss += kGeoBlock) {
            long long vol_off = 0, img_off = 0, r = ss;
            for (int l = 0; l < hg.nstep; ++l) {
                const long long q = r / hg.step_len[l];
                const long long c = r - q * hg.step_len[l];
                vol_off += hg.vol_step[l] * c;
                img_off += hg.img_step[l] * c;
                r = q;
            }
            zg.steps[2 * ss] = vol_off;
            zg.steps[2 * ss + 1] = img_off;
        }
    }
    if (blockIdx.x == 0 && sample == 0 && tid == 0) {
        ZGen* zn = reinterpret_cast<ZGen*>(zg.zgen);
        for (int h = 0; h < 3; ++h) {
            zn->in_len[h] = hg.in_len[h];
            zn->out_len[h] = hg.out_len[h];
            zn->off[h] = hg.off[h];
            zn->period[h] = hg.period[h];
            zn->inv_period[h] = hg.inv_period[h];
            zn->offd[h] = (double)hg.off[h];
        }
        for (int k = 0; k < 12; ++k)
            zn->aff[k] = hg.affine[k];
        zn->mode = hg.mode;
        zn->cval = hg.cval;
        zn->hint = hg.hint;
    }
    if (blockIdx.x == 0 && sample == 0) {
        AxTab* zt = const_cast<AxTab*>(zg.zt);
        for (int oz = tid; oz < hg.out_len[0]; oz += kGeoBlock) {
            AxTab t;
            zaxis_entry(g, 0, oz, t.w, t.idx);
#pragma unroll
            for (int l = 0; l < 4; ++l)
                t.idx[l] *= 32;              // byte offset of the control plane in an R column [ncp_z][4 doubles]
            zt[oz] = t;
        }
    }
    __syncthreads();
    ZTICK(3);
    double slack[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        double m = 0.0;
        for (int w = 0; w < kGeoWaves; ++w)
            m = fmax(m, sMax[h * kGeoWaves + w]);
        const double sl = zg.slack_scale * m;
        slack[h] = sl >= 0.02 ? (sl <= 0.75 ? sl : 0.75) : 0.02;        // (NaN -> 0.02)
    }
    ZTICK(4);
    // ---- R of the patch: contraction over y and x (deform.c:693-758, two of its three axes) --------
    // A thread keeps ONE column's axis entries in registers and computes every fourth (k_z, component) of it (the
    // entries used to be re-read from LDS per output); the patch's R leaves through LDS in rows of 8 columns x ncp_z x 4
    // doubles, which are contiguous in the global layout (it left as 960 scattered 8-byte stores per workgroup).
    if (!ED_DBG(hg.dbg, 1 << 25)) {
        const int col = tid & 63;
        const AxTab ay = sAx[col >> 3];
        const AxTab ax = sAx[8 + (col & 7)];
        int rowoff[4];
#pragma unroll
        for (int ly = 0; ly < 4; ++ly)
            rowoff[ly] = ay.idx[ly] * ncpx;
        for (int r = tid >> 6; r < ncpz * 3; r += kGeoWaves) {
            const int kz = r / 3, h = r - kz * 3;
            const double* gp0 = sG + (h * ncpz + kz) * nyx;
            double acc = 0.0;
#pragma unroll
            for (int ly = 0; ly < 4; ++ly) {
                const double* row = gp0 + rowoff[ly];
                double t = ax.w[0] * row[ax.idx[0]];
#pragma unroll
                for (int lx = 1; lx < 4; ++lx)
                    t = fma(ax.w[lx], row[ax.idx[lx]], t);
                acc = fma(ay.w[ly], t, acc);
            }
            sR[col * (ncpz * 3) + r] = acc;
        }
        __syncthreads();
        double* rg = const_cast<double*>(zg.r) + (int64_t)sample * zg.r_bstride;
        const int rowlen = 8 * 4 * ncpz;                      // doubles of a row of 8 columns in the global layout
        for (int e = tid; e < 8 * rowlen; e += kGeoBlock) {
            const int py = e / rowlen, j = e - py * rowlen;
            const int px = j / (4 * ncpz), jj = j - px * (4 * ncpz);
            const int kz = jj >> 2, h = jj & 3;
            const int oy = ty * kT + py, ox = tx * kT + px;
            if (oy < hg.out_len[1] && ox < hg.out_len[2])
                rg[((int64_t)oy * hg.out_len[2] + ox) * (4 * ncpz) + jj] = h < 3 ? sR[(py * 8 + px) * (ncpz * 3) + kz * 3 + h] : 0.0;
        }
    }
    if (zg.tables_only)
        return;          // (the gradient route: R, the z table and ZGen are all it reads)
    __syncthreads();
    ZTICK(5);
    // ---- tile boxes of the patch's z column: wave w samples tiles w, w + 8, ...; lane 63 leaves the reduced ranges in LDS
    const int order = zg.order;
    const int H = order / 2, NT = order + 1, kPadX = NT & 1;
    const int ny = min(kT, hg.out_len[1] - ty * kT), nx = min(kT, hg.out_len[2] - tx * kT);
    const int py = (((lane >> 2) & 3) * (ny - 1)) / 3, px = ((lane & 3) * (nx - 1)) / 3;
    const double* rc = sR + (py * 8 + px) * (ncpz * 3);
    for (int tz = wave; tz < hg.tiles[0] && !ED_DBG(hg.dbg, 1 << 24); tz += kGeoWaves) {
        const int nz = min(kT, hg.out_len[0] - tz * kT);
        const int pz = ((lane >> 4) * (nz - 1)) / 3;
        const int o[3] = {tz * kT + pz, ty * kT + py, tx * kT + px};
        const AxTab& ze = sZ[tz * 4 + (lane >> 4)];
        double d[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            d[h] = ze.w[0] * rc[ze.idx[0] * 3 + h];
#pragma unroll
            for (int l = 1; l < 4; ++l)
                d[h] = fma(ze.w[l], rc[ze.idx[l] * 3 + h], d[h]);
        }
        double c[3];
        int lo[3], hi[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            if (hg.has_affine)
                c[h] = fma(hg.affine[h * 4 + 2], (double)o[2],
                           fma(hg.affine[h * 4 + 0], (double)o[0],
                               fma(hg.affine[h * 4 + 1], (double)o[1], hg.affine[h * 4 + 3] + (double)hg.off[h]))) + d[h];
            else
                c[h] = (double)(o[h] + hg.off[h]) + d[h];
            const double cr = (order & 1) ? c[h] : c[h] + 0.5;
            lo[h] = (int)floor(cr - slack[h]);
            hi[h] = (int)floor(cr + slack[h]);
        }
        // fast: a full tile whose every coordinate, margin included, is one coord_axis_fast accepts
        bool out_of_fast = false;
#pragma unroll
        for (int h = 0; h < 3; ++h)
            out_of_fast = out_of_fast || lo[h] < ((order & 1) ? 0 : 1) || hi[h] > hg.in_len[h] - 2;
        const bool fast = nz == kT && ny == kT && nx == kT && !__any(out_of_fast);
        // every lane of a row of 16 holds the row's range = the range of one sampled z slice; the thread that derives
        // the tile's record combines the slices (whole tile, low half, high half)
        zrow_box(lo, hi);
        int* dst = sBox + (tz * 8 + (lane >> 4)) * 8;
        if ((lane & 15) == 0) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                dst[h] = lo[h];
                dst[4 + h] = hi[h];
            }
        }
        if (lane == 0)
            dst[3] = fast ? 1 : 0;
        if (!fast) {
            // general tile: the range of the MAPPED coordinate over the samples (a sample that maps to the constant has
            // no window); where the raw range straddles an end of the array, the end itself is included below
            bool cst = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                double m = c[h];
                if (!(m >= 0.0 && m <= (double)(hg.in_len[h] - 1)))
                    m = map_coordinate_fast(m, hg.in_len[h], hg.mode, hg.period[h], hg.inv_period[h]);
                cst = cst || !(m > -1.0);
                const double mr = (order & 1) ? m : m + 0.5;
                lo[h] = (int)floor(mr - slack[h]);
                hi[h] = (int)floor(mr + slack[h]);
            }
            if (cst) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    lo[h] = 0x7fffffff;
                    hi[h] = (int)0x80000000;
                }
            }
            zrow_box(lo, hi);
            if ((lane & 15) == 0) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    dst[32 + h] = lo[h];
                    dst[36 + h] = hi[h];
                }
            }
        }
    }
    __syncthreads();
    ZTICK(6);
    // ---- tile records: one THREAD per tile of the column (the derivation is a chain of ~100 scalar steps: as lane 63
    //      of the sampling wave it cost every tile a wave's issue slots) --------------------------------------------
    for (int tz = tid; tz < hg.tiles[0] && !ED_DBG(hg.dbg, 1 << 27); tz += kGeoBlock) {
        const int tile_id = sample * hg.ntiles + (tz * hg.tiles[1] + ty) * hg.tiles[2] + tx;
        int cls = 0;
        bool hint0 = false, half_fit[2] = {false, false};
        // part 0: the whole tile; parts 1 / 2: its halves along z (the lane's first / second voxel), used when the whole
        // tile's box does not fit LDS
#pragma unroll 1
        for (int part = 0; part < 3; ++part) {
            // rows = the sampled z slices 0, 2, 4, 7: the whole tile; its low half (slices 0-3: the samples at 0, 2 and 4 --
            // the sample at 4 bounds slice 3 by interpolation); its high half (slices 4-7: the samples at 4 and 7)
            const int* sb = sBox + tz * 64;
            const bool fast = sb[3] != 0;
            const int r0 = part == 2 ? 2 : 0, r1 = part == 1 ? 2 : 3;
            int blo[3], bhi[3];
            bool any = true;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                int rlo = 0x7fffffff, rhi = (int)0x80000000, lo = 0x7fffffff, hi = (int)0x80000000;
                for (int r = r0; r <= r1; ++r) {
                    rlo = min(rlo, sb[r * 8 + h]);
                    rhi = max(rhi, sb[r * 8 + 4 + h]);
                    lo = min(lo, sb[(fast ? 0 : 32) + r * 8 + h]);
                    hi = max(hi, sb[(fast ? 0 : 32) + r * 8 + 4 + h]);
                }
                if (!fast) {
                    any = any && hi >= lo;
                    // the lowest / highest floor a coordinate that stays in (or is folded / clamped back into) the array
                    // can have, wherever the raw range reaches beyond what coord_axis_fast accepts
                    const int lowfold = hg.mode == EDHIP_MODE_REFLECT ? -1 : 0;
                    if (rlo < ((order & 1) ? 0 : 1) && rhi >= lowfold)
                        lo = min(lo, lowfold);
                    if (rhi > hg.in_len[h] - 2 && rlo <= hg.in_len[h] - 1)
                        hi = max(hi, hg.in_len[h] - 1);
                }
                blo[h] = lo - H;
                bhi[h] = hi - H + order + (h == 2 ? kPadX : 0);
            }
            if (!any) {          // every sample maps to the constant
                blo[0] = blo[1] = blo[2] = 0;
                bhi[0] = bhi[1] = bhi[2] = -1;
            }
            any = bhi[0] >= blo[0] && bhi[1] >= blo[1] && bhi[2] >= blo[2];
            unsigned ext[3];
#pragma unroll
            for (int h = 0; h < 3; ++h)
                ext[h] = (unsigned)bhi[h] - (unsigned)blo[h] + 1u;
            // floats per box row: 16 for the mild fields; 24 / 32 / 48 where the deformation stretches a tile along x
            // (every pitch a multiple of 8 floats: the pairs of the shifted copy stay aligned)
            const int pitch = ext[2] <= 16u ? 16 : (ext[2] <= 24u ? 24 : (ext[2] <= 32u ? 32 : (ext[2] <= 48u ? 48 : 0)));
            const bool sane = any && ext[0] < 1024u && ext[1] < 1024u;
            const int nrows = sane ? (int)(ext[0] * ext[1]) : 0;
            const bool fits = sane && pitch > 0 && nrows * pitch <= hg.box_cap;
            const bool dma = fits && blo[2] >= 0 && blo[2] + pitch + 1 <= hg.in_len[2];
            const bool xin = fits && blo[2] >= 0 && blo[2] + (int)ext[2] <= hg.in_len[2];
            const bool zyin = fits && blo[0] >= 0 && blo[0] + (int)ext[0] <= hg.in_len[0] && blo[1] >= 0 &&
                              blo[1] + (int)ext[1] <= hg.in_len[1];
            // class A (k1z_fast_kernel): full tile, coordinates inside the array, box inside the array, LDS-DMA rows.
            // Everything else that fits is a general tile (k1z_gen_kernel); a tile that is fast but whose box touches
            // the array's ends has raw == mapped coordinates, so its box is a general tile's box as it stands.
            const bool fastA = part == 0 && fast && fits && dma && zyin;
            const int flags = (any ? kZAny : 0) | (fits ? kZStaged : 0) | (fastA ? kZFast : kZGen) | (dma ? kZDma : 0) |
                              (xin ? kZXin : 0) | (zyin ? kZZYin : 0) | (any && !fits ? kZUnfit : 0);
            const int4 w0 = make_int4(blo[0], blo[1], blo[2], flags | (pitch << 8));
            const int4 w1 = make_int4(sane ? (int)(ext[0] | (ext[1] << 10) | (ext[2] << 20)) : 0,
                                      (dma && zyin) ? blo[0] * hg.vol_sz + blo[1] * hg.vol_sy + blo[2] : 0, 0, 0);
            if (part == 0) {
                hint0 = any && !(sane && pitch > 0 && nrows * pitch <= hg.small_cap);
                cls = (flags & kZUnfit) ? 3 : ((flags & kZFast) ? 1 : 2);
                int4* dst = reinterpret_cast<int4*>(zg.recs + (size_t)tile_id * 8);
                dst[0] = w0;
                dst[1] = w1;
                if (hg.boxes) {
                    int* bx = hg.boxes + (size_t)tile_id * 8;
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        bx[h] = blo[h];
                        bx[3 + h] = bhi[h] - ((h == 2 && any) ? kPadX : 0);      // (without the forward gather's padding tap)
                    }
                }
                if (cls != 3)
                    break;               // (the halves matter only where the whole tile does not fit)
            } else {
                // a half without windows (every sample constant) "fits": its voxels store the constant
                half_fit[part - 1] = fits || !any;
                int4* dst = reinterpret_cast<int4*>(zg.recs_half + ((size_t)tile_id * 2 + (part - 1)) * 8);
                dst[0] = w0;
                dst[1] = w1;
            }
        }
        const bool split = cls == 3 && half_fit[0] && half_fit[1];
        ZSTAT(cls == 3 ? (split ? 6 : 5) : (cls == 1 ? 3 : 4), 1);
        // class of the tile for the strip summary: 1 class A, 2 general, 3 does not fit; + 4: does not fit but its halves
        // do (taken by the general kernel half by half); + 8: beyond the standard box
        sCls[tz] = cls | (split ? 4 : 0) | (hint0 ? 8 : 0);
    }
    __syncthreads();
    ZTICK(7);
    // ---- strip summaries (4 bits per tile: class, halves fit, beyond the standard box), the "missed" flags, the work lists:
    //      G = strips with general tiles (k1z_gen_kernel), F = strips with tiles that do not fit (k1z_fix_kernel; the
    //      tile kernels append the strips in which a window fell outside its sampled box).  One atomic per list and
    //      column.  The counters of THIS call (parity p) were cleared by the previous call's geometry kernel; this one
    //      clears the next call's. -----------------------------------------------------------------------------------
    {
        const int zstrips = (hg.tiles[0] + zg.strip_tiles - 1) / zg.strip_tiles;
        if (tid < 4)
            sCnt[tid] = 0;
        __syncthreads();
        int slotG = -1, slotF = -1, sidv = 0;
        if (tid < zstrips) {
            int info = 0;
            bool g = false, f = false;
            for (int k = 0; k < zg.strip_tiles && tid * zg.strip_tiles + k < hg.tiles[0]; ++k) {
                const int c = sCls[tid * zg.strip_tiles + k];
                info |= c << (4 * k);
                g = g || (c & 3) == 2 || (c & 7) == 7;
                f = f || (c & 7) == 3;
            }
            const size_t sid = (size_t)sample * zg.nstrips + ((size_t)tid * hg.tiles[1] + ty) * hg.tiles[2] + tx;
            sidv = (int)sid;
            zg.sinfo[sid] = info;
            zg.missed[sid] = f ? 2 : 0;            // (2: already on list F)
            if (g)
                slotG = atomicAdd(&sCnt[0], 1);
            if (f)
                slotF = atomicAdd(&sCnt[1], 1);
        }
        __syncthreads();
        // (eight counters per list, one per XCD: block b runs on XCD b % 8; all the workgroups of this launch arrive here
        // at about the same time, and a thousand returning atomics on ONE word took 24 us)
        // (skewed: a face of the volume -- tx = 0, or ty = 0 -- must not land on one XCD's list)
        const int xcd = (tx + ty + sample) & 7;
        if (tid == 0) {
            sCnt[2] = sCnt[0] ? atomicAdd(&zg.ctl[zctl(zg.parity, 0, xcd)], sCnt[0]) : 0;
            sCnt[3] = sCnt[1] ? atomicAdd(&zg.ctl[zctl(zg.parity, 1, xcd)], sCnt[1]) : 0;
        }
        if (blockIdx.x == 0 && sample == 0 && tid < 16)
            zg.ctl[zctl(1 - zg.parity, tid >> 3, tid & 7)] = 0;
        __syncthreads();
        if (slotG >= 0)
            zg.list_g[(size_t)(sCnt[2] + slotG) * 8 + xcd] = sidv;
        if (slotF >= 0)
            zg.list_f[(size_t)(sCnt[3] + slotF) * 8 + xcd] = sidv;
    }
    ZTICK(8);
#undef ZTICK
}

// ================================================================================================
// the forward kernels
// ================================================================================================
// Three launches per call behind the geometry kernel:
//   k1z_fast_kernel   class-A tiles (full tile, coordinates inside the array, box inside the array, fits): ~89 % of the
//                     benchmark's tiles.  No boundary tests, one staging path, a small argument block: nothing but the
//                     per-voxel work is left in the loop.
//   k1z_gen_kernel    general tiles (array faces, partial tiles, boxes that touch the array's ends), on a second stream
//                     next to the fast kernel, persistent over the list of strips that have such tiles.
//   k1z_fix_kernel    behind both: tiles whose box does not fit LDS, and the voxels of staged tiles whose window was not
//                     inside the sampled box (the tile kernels raise a flag per strip), straight from global memory.
// All walk the same strips; a voxel gets the same bits whichever serves it (same R, same z table, same sums).

// per-voxel state handed from the coordinate pass of tile t + 1 to its gather, one tile later
struct ZVox {
    int addr[2];          // LDS byte address of tap (0, 0, 0) in the copy that matches the window's parity
    float frac[2][3];
    int flg;              // general tiles: bit i = voxel i is gathered, bit 2 + i = voxel i is stored
};
struct ZStrip {
    int tz0, ty, tx, ntile, sample, id;
};
// Strips (runs of tiles along z at one (ty, tx)) are dealt to the 8 XCDs (block b runs on XCD b % 8) in chunks of `deal`
// consecutive strips -- strips next to each other in x and y, whose source boxes overlap, share an L2 -- and the chunks
// go round the XCDs with a skew of one per round, so that no XCD collects a face of the volume (the tiles there are
// general tiles: with one contiguous range per XCD the two XCDs that owned the z faces had a quarter of the class-A work
// and all of the rest kernel's).  deal 1: plain round robin.
__device__ __forceinline__ bool k1z_strip(int total_strips, int nstrips, int tiles_z, int tiles_y, int tiles_x, int strip_tiles, int deal,
                                          ZStrip& sp, int b)
{
    const int x = b & 7, j = b >> 3;
    const int r = j / deal;
    int s = (r * 8 + ((x + r) & 7)) * deal + (j - r * deal);
    if (s >= total_strips)
        return false;
    sp.id = s;
    sp.sample = s / nstrips;
    s -= sp.sample * nstrips;
    sp.tx = s % tiles_x;
    s /= tiles_x;
    sp.ty = s % tiles_y;
    sp.tz0 = (s / tiles_y) * strip_tiles;
    sp.ntile = min(strip_tiles, tiles_z - sp.tz0);
    return true;
}
// blocks a launch needs so that every strip is dealt
inline unsigned k1z_grid(int total_strips, int deal)
{
    const int chunks = (total_strips + deal - 1) / deal;
    return (unsigned)(((chunks + 7) / 8) * 8 * deal);
}

// 64-tap (order 3) separable gather of one voxel from the staged box (see deform_k1.hip: reads kept apart, aligned
// 8-byte pairs from the copy that matches the window's parity)
template <int ORDER, int PITCH>
__device__ __forceinline__ float k1z_gather(const float* bp, int plane, int pitch_rt, const float* w0, const float* w1, const float* w2)
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
            const float* rp = pp + l1 * (PITCH ? PITCH : pitch_rt);
            float a2 = 0.f;
#pragma unroll
            for (int l2 = 0; l2 < NTX; l2 += 2) {
                const float2 pr = *reinterpret_cast<const float2*>(rp + l2);
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
// one voxel from the staged box: weights from the fractions, then the gather
template <int ORDER>
__device__ __forceinline__ float k1z_voxel(const char* smem, int addr, const float (&frac)[3], int pitch, int plane)
{
    constexpr int NT = ORDER + 1;
    constexpr int NTX = NT + (NT & 1);
    // (the fractions go through an empty asm statement: left alone, the compiler computes the weights of both
    // voxels in front of the first gather and holds 24 registers across it)
    float f0 = frac[0], f1 = frac[1], f2 = frac[2];
    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2));
    float w0[NT], w1[NT], w2[NTX];
    weights_from_frac<float, ORDER>(f0, w0);
    weights_from_frac<float, ORDER>(f1, w1);
    weights_from_frac<float, ORDER>(f2, w2);
    if (NT & 1)
        w2[NT] = 0.f;
    const float* bp = reinterpret_cast<const float*>(smem + addr);
    // (PITCH 0: the row offsets of the rare wide boxes are computed, not immediates)
    return pitch == 16 ? k1z_gather<ORDER, 16>(bp, plane, 16, w0, w1, w2) : k1z_gather<ORDER, 0>(bp, plane, pitch, w0, w1, w2);
}

// ---- class-A tiles ---------------------------------------------------------------------------------------------
struct ZFast {
    const float* vol;
    float* img;
    const double* r;
    const AxTab* zt;
    const int* recs;
    const int* recs_half;     // [tile][2][8]: z halves of the tiles whose whole box does not fit
    int* missed;              // [strip]: bit 0: a window of a class-A tile was not inside its sampled box
    int* hint;                // spill feedback: count of the tiles that do not fit the standard box (or nullptr)
    const int* sinfo;         // [strip]: tile classes (geometry kernel)
    const int* list_g;        // strips with general tiles (geometry kernel); count in ctl[parity]
    int* list_f;              // strips for the fix-up kernel; count in ctl[2 + parity]
    int* ctl;
    int parity;
    const long long* steps;   // STEPS: [nsteps][2] element offsets (volume, image) of the step axes' indices
    long long vol_bstride, img_bstride, r_bstride;
    int vol_sy, vol_sz, img_sy, img_sz;
    int box_cap, tiles_y, tiles_x, tiles_z, strip_tiles, nstrips, total_strips, ntiles;
    int rcol_bytes, out_y, out_x;
    int kz0, ky0, kx0;        // window start = floor part of the coordinate + k: crop offset - order / 2 (affine: - order / 2)
    int io16, nsteps, deal;
    int dbg;                  // profiling build: ablation bits (1 << 17 no gather, 1 << 18 no staging, 1 << 20 no stores)
    double aff[12];           // AFFINE: inverse map, the crop offset folded into column 3
};

template <int ORDER, bool AFFINE, bool OUT16, bool STEPS>
__device__ __forceinline__ void k1z_fast_body(const ZFast& a, const int vblock)
{
    constexpr int NT = ORDER + 1;
    constexpr int kPadX = NT & 1;          // even orders read one zero-weight padding tap
    constexpr int NTX = NT + kPadX;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the box pair, nothing else
    ZStrip sp;
    if (!k1z_strip(a.total_strips, a.nstrips, a.tiles_z, a.tiles_y, a.tiles_x, a.strip_tiles, a.deal, sp, vblock))
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = zuni(tid >> 6);
    const int io16 = OUT16 ? a.io16 : 0;
    const int ntile = sp.ntile;
    const size_t tile_step = (size_t)a.tiles_y * a.tiles_x;      // records between tz and tz + 1
    crec_p rec0 = (crec_p)(const void*)a.recs + ((size_t)sp.sample * a.ntiles + ((size_t)sp.tz0 * a.tiles_y + sp.ty) * a.tiles_x + sp.tx);
    // the strip's summary (geometry kernel): 3 bits per tile -- class 1 = A, 2 = general, 3 = does not fit
    const int sinfo = ((cint_p)(const void*)a.sinfo)[sp.id];
    if (a.hint && tid == 0) {
        // spill feedback: the strip's tiles that do not fit the standard box (this kernel visits every strip)
        const int nh = __builtin_popcount((unsigned)sinfo & 0x88888888u);
        if (nh)
            atomicAdd(a.hint, nh);
    }
    auto next_fast = [&](int t) {
        for (++t; t < ntile; ++t)
            if (((sinfo >> (4 * t)) & 3) == 1)
                break;
        return t;
    };
    int ti = next_fast(-1);
    if (ti >= ntile)
        return;                            // (no class-A tile in this strip)

    const int box_cap = a.box_cap;
    const int odd_shift = (box_cap - 1) * 4;               // first copy -> second copy, one element back
    float* box0 = reinterpret_cast<float*>(smem);
    float* box1 = box0 + box_cap;          // cap = 56 (mod 64): the two copies sit on disjoint banks
    const int vol_sz = a.vol_sz, vol_sy = a.vol_sy, img_sz = a.img_sz;

    // per-lane values that stay fixed along the strip
    const int yy = lane >> 3, xx = lane & 7;
    const int oy = sp.ty * kT + yy, ox = sp.tx * kT + xx;
    const int obase = oy * a.img_sy + ox;
    const float* __restrict__ vol = a.vol + sp.sample * a.vol_bstride;
    float* img = a.img + sp.sample * a.img_bstride;
    const char* rcol = reinterpret_cast<const char*>(a.r + sp.sample * a.r_bstride) +
                       ((size_t)min(oy, a.out_y - 1) * a.out_x + min(ox, a.out_x - 1)) * (size_t)a.rcol_bytes;
    const int ky = AFFINE ? a.ky0 : oy + a.ky0;
    const int kx = AFFINE ? a.kx0 : ox + a.kx0;
    const int kz0 = a.kz0;
    double Pyx[3];       // affine: A[h][1] oy + A[h][2] ox + A[h][3] (+ off_h)
#pragma unroll
    for (int h = 0; h < 3; ++h)
        Pyx[h] = AFFINE ? fma(a.aff[h * 4 + 2], (double)ox, fma(a.aff[h * 4 + 1], (double)oy, a.aff[h * 4 + 3])) : 0.0;
    cdbl_p zt = (cdbl_p)(const void*)a.zt;
    // staging: lane -> (row of a plane, 16-byte chunk of the row); 16 rows of 64 bytes or 5 rows of 192 bytes per KiB
    const int lr16 = lane >> 2, q16 = lane & 3;

    ZTaps tp;
    tp.key[0] = tp.key[1] = tp.key[2] = tp.key[3] = -1;

    // the source box of a tile into LDS, two copies, the second shifted by one element: LDS-DMA, plane by plane (planes
    // dealt to the four waves), one wave-instruction fills 1 KiB
    auto stage = [&](const ZRecU& rc, const float* src) {
        if (rc.pitch == 16) {
            const float* g0 = src + (rc.goff + lr16 * vol_sy + 4 * q16);
            for (int zrow = wave; zrow < rc.ez; zrow += 4) {
                const float* gp = g0 + zrow * vol_sz;
                const int lrow0 = zrow * rc.ey;
                for (int y0 = 0; y0 < rc.ey; y0 += 16) {
                    if (y0 + lr16 < rc.ey) {
                        const float* g = gp + y0 * vol_sy;
                        zglds16(g, box0 + (lrow0 + y0) * 16);
                        zglds16(g + 1, box1 + (lrow0 + y0) * 16);
                    }
                }
            }
        } else {
            // wider rows: 6 / 8 / 12 chunks of 16 bytes, 10 / 8 / 5 rows per wave-instruction
            const int cpr = rc.pitch >> 2, RW = 64 / cpr;
            const int lr = (lane * (65536 / cpr + 1)) >> 16, q = lane - lr * cpr;
            const float* g0 = src + (rc.goff + lr * vol_sy + 4 * q);
            for (int zrow = wave; zrow < rc.ez; zrow += 4) {
                const float* gp = g0 + zrow * vol_sz;
                const int lrow0 = zrow * rc.ey;
                for (int y0 = 0; y0 < rc.ey; y0 += RW) {
                    if (lr < RW && y0 + lr < rc.ey) {
                        const float* g = gp + y0 * vol_sy;
                        zglds16(g, box0 + (lrow0 + y0) * rc.pitch);
                        zglds16(g + 1, box1 + (lrow0 + y0) * rc.pitch);
                    }
                }
            }
        }
    };
    // coordinates of the lane's two voxels of a tile, as LDS addresses relative to the tile's box; a window that is not
    // inside the box raises the lane's flag
    int bad = 0;
    auto tile_coords = [&](int t, const ZRecU& rc, const ZEnt (&ze)[2], ZVox& vs) {
        const int ez = rc.ez - NT, ey = rc.ey - NT, ex = rc.ex - NTX;
        const int dy = ky - rc.b0y, dx = kx - rc.b0x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oz = (sp.tz0 + t) * kT + wave + 4 * i;
            k1z_taps(ze[i], rcol, tp);
            const double zw[4] = {ze[i].w[0], ze[i].w[1], ze[i].w[2], ze[i].w[3]};
            double d[3];
            k1z_disp(tp, zw, d);
            int ci[3];
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                // (coord_axis_fast without its range test: the same floor, the same fraction)
                const double c = AFFINE ? fma(a.aff[h * 4 + 0], (double)oz, Pyx[h]) + d[h] : d[h];
                const double fl = floor((ORDER & 1) ? c : c + 0.5);
                ci[h] = (int)fl;
                vs.frac[i][h] = (float)(c - fl);
            }
            const int rz = ci[0] + ((AFFINE ? 0 : oz) + kz0 - rc.b0z);
            const int ry = ci[1] + dy;
            const int rx = ci[2] + dx;
            bad |= (rz | ry | rx) | ((ez - rz) | (ey - ry) | (ex - rx));
            // (24-bit multiplies: a window inside the box has small non-negative offsets; one outside is flagged)
            const int off = __mul24(rz, rc.plane) + (__mul24(ry, rc.pitch) + rx);
            // aligned pairs from the copy whose shift matches the parity of rx (pitch and plane are even)
            vs.addr[i] = __mul24(off & 1, odd_shift) + off * 4;
        }
    };

    // ---- tile loop, software-pipelined: the coordinates of the next class-A tile under the copies of this one ------
    cll_p steps = (cll_p)(const void*)a.steps;
    ZVox cur, nxt;
    ZRecU rc = zrec_load(rec0 + (size_t)ti * tile_step), rn = rc;
    {
        const ZEnt ze[2] = {k1z_entry(zt, (sp.tz0 + ti) * kT + wave), k1z_entry(zt, (sp.tz0 + ti) * kT + wave + 4)};
        tile_coords(ti, rc, ze, cur);
    }
    while (ti < ntile) {
        // the next tile's record and z-table entries: scalar loads issued here, ahead of the staging loop, used behind it
        const int tn = next_fast(ti);
        const int tl = tn < ntile ? tn : ti;
        const zv8i rnv = *(rec0 + (size_t)tl * tile_step);
        const ZEnt zn[2] = {k1z_entry(zt, (sp.tz0 + tl) * kT + wave), k1z_entry(zt, (sp.tz0 + tl) * kT + wave + 4)};
        if (!ED_DBG(a.dbg, 1 << 18))
            stage(rc, vol + (STEPS ? steps[0] : 0));
        if (tn < ntile) {
            rn = zrec_unpack(rnv);
            tile_coords(tn, rn, zn, nxt);
        }
        const long long ozoff = (long long)((sp.tz0 + ti) * kT + wave) * img_sz + obase;
        const int nsteps = STEPS ? a.nsteps : 1;
        for (int ss = 0; ss < nsteps; ++ss) {
            if (STEPS && ss > 0)
                stage(rc, vol + steps[2 * ss]);
            zdma_barrier();       // B2: retires this wave's copies (vmcnt) and everyone's
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float val = ED_DBG(a.dbg, 1 << 17) ? cur.frac[i][0] + cur.frac[i][1] + cur.frac[i][2] + __int_as_float(cur.addr[i])
                                                         : k1z_voxel<ORDER>(smem, cur.addr[i], cur.frac[i], rc.pitch, rc.plane);
                // streaming store (a tile writes 32-byte row segments)
                if (!ED_DBG(a.dbg, 1 << 20) || val == 123.456f)
                    zstore_out<OUT16>(img, (STEPS ? steps[2 * ss + 1] : 0) + ozoff + (long long)(4 * i) * img_sz, val, io16);
            }
            zlds_barrier();       // B1: every gather of this tile is done with the box
        }
        cur = nxt;
        rc = rn;
        ti = tn;
    }
    if (__any(bad < 0) && lane == 0) {
        // the fix-up kernel redoes the voxels of this strip whose window is not inside the box
        if (atomicOr(a.missed + sp.id, 1) == 0) {
            const int xcd = vblock & 7;
            a.list_f[(size_t)atomicAdd(a.ctl + zctl(a.parity, 1, xcd), 1) * 8 + xcd] = sp.id;
        }
    }
}

// ---- everything else ---------------------------------------------------------------------------------------------
// General tiles (array faces, partial tiles, boxes that touch the array's ends): general coordinates (deform.c:771-824)
// with the boundary map, constant and valid flags, every staging path.  Persistent over list G (the strips with such
// tiles); launched on a second stream next to the fast kernel, so that its long strips -- the columns on the x faces,
// whose boxes are staged element by element -- overlap with the class-A work instead of trailing it.  Same walk, same R
// and z table, same sums as the fast kernel: the same bits.
template <int ORDER, bool AFFINE, bool OUT16, bool STEPS>
__device__ __forceinline__ void k1z_gen_body(const ZFast& a, czgen_p zn, const int vblock, const int ngen)
{
    constexpr int NT = ORDER + 1;
    constexpr int kPadX = NT & 1;
    constexpr int NTX = NT + kPadX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = zuni(tid >> 6);
    const int io16 = OUT16 ? a.io16 : 0;
    const int xcd = vblock & 7, nper = ngen >> 3;
    const int nwork = ((cint_p)(const void*)a.ctl)[zctl(a.parity, 0, xcd)];      // (this XCD's share of list G)
    if ((vblock >> 3) >= nwork)
        return;
    const int box_cap = a.box_cap;
    const int odd_shift = (box_cap - 1) * 4;
    float* box0 = reinterpret_cast<float*>(smem);
    float* box1 = box0 + box_cap;
    const int vol_sz = a.vol_sz, vol_sy = a.vol_sy, img_sz = a.img_sz;
    const size_t tile_step = (size_t)a.tiles_y * a.tiles_x;
    cdbl_p zt = (cdbl_p)(const void*)a.zt;
    cll_p steps = (cll_p)(const void*)a.steps;
    const int nsteps = STEPS ? a.nsteps : 1;
    const int yy = lane >> 3, xx = lane & 7;

    for (int work = vblock >> 3; work < nwork; work += nper) {
        ZStrip sp;
        {
            int sid = ((cint_p)(const void*)a.list_g)[(size_t)work * 8 + xcd];
            sp.id = sid;
            sp.sample = sid / a.nstrips;
            sid -= sp.sample * a.nstrips;
            sp.tx = sid % a.tiles_x;
            sid /= a.tiles_x;
            sp.ty = sid % a.tiles_y;
            sp.tz0 = (sid / a.tiles_y) * a.strip_tiles;
            sp.ntile = min(a.strip_tiles, a.tiles_z - sp.tz0);
        }
        const int ntile = sp.ntile;
        crec_p rec0 = (crec_p)(const void*)a.recs + ((size_t)sp.sample * a.ntiles + ((size_t)sp.tz0 * a.tiles_y + sp.ty) * a.tiles_x + sp.tx);
        const int sinfo = ((cint_p)(const void*)a.sinfo)[sp.id];
        crec_p rech0 = (crec_p)(const void*)a.recs_half + ((size_t)sp.sample * a.ntiles + ((size_t)sp.tz0 * a.tiles_y + sp.ty) * a.tiles_x + sp.tx) * 2;
        if (tid == 0)
            ZSTAT(7, 1);
        if (work != (vblock >> 3))
            zlds_barrier();                        // (the previous strip's gathers are done with the box)
        const int oy = sp.ty * kT + yy, ox = sp.tx * kT + xx;
        const int obase = oy * a.img_sy + ox;
        const float* __restrict__ vol = a.vol + sp.sample * a.vol_bstride;
        float* img = a.img + sp.sample * a.img_bstride;
        const char* rcol = reinterpret_cast<const char*>(a.r + sp.sample * a.r_bstride) +
                           ((size_t)min(oy, a.out_y - 1) * a.out_x + min(ox, a.out_x - 1)) * (size_t)a.rcol_bytes;
        double Pyx[3];
#pragma unroll
        for (int h = 0; h < 3; ++h)
            Pyx[h] = AFFINE ? fma(zn->aff[h * 4 + 2], (double)ox, fma(zn->aff[h * 4 + 1], (double)oy, zn->aff[h * 4 + 3] + zn->offd[h])) : 0.0;
        ZTaps tp;
        tp.key[0] = tp.key[1] = tp.key[2] = tp.key[3] = -1;
        // ---- staging: the source box of a tile into LDS, two copies, the second shifted by one element ----------
        auto stage = [&](const ZRecU& rc, const float* src) {
            const int flags = rc.flags;
            if (!(flags & kZStaged))
                return;
            const int by = rc.ey, pitch = rc.pitch;
            if (flags & kZDma) {
                const int cpr = pitch >> 2, RW = 64 / cpr;                  // 16-byte chunks per row, rows per KiB
                const int lr = (lane * (65536 / cpr + 1)) >> 16, q = lane - lr * cpr;      // lane / cpr
                // planes / rows beyond the array's z / y ends: the mirror map of the reference's taps
                // (deform.c:791-813), applied to the plane / row index
                const float* g0 = src + (rc.b0x + 4 * q);
                for (int zrow = wave; zrow < rc.ez; zrow += 4) {
                    const float* gp = g0 + (long long)mirror_i32(rc.b0z + zrow, zn->in_len[0]) * vol_sz;
                    const int lrow0 = zrow * by;
                    for (int y0 = 0; y0 < by; y0 += RW) {
                        if (lr < RW && y0 + lr < by) {
                            const float* g = gp + (long long)mirror_i32(rc.b0y + y0 + lr, zn->in_len[1]) * vol_sy;
                            zglds16(g, box0 + (lrow0 + y0) * pitch);
                            zglds16(g + 1, box1 + (lrow0 + y0) * pitch);
                        }
                    }
                }
            } else {
                // the box sticks out along x: every box index goes through the mirror map, as the reference does with
                // the taps of a window that sticks out (deform.c:791-813)
                const int nrows = rc.ez * rc.ey;
                const float inv_by = __frcp_rn((float)by);
                const int sub = tid & 7;
                // (two rows = four loads of a thread in flight: one element at a time every load waited for the one before -- the boxes of
                // the columns on the x faces took ~10 us each)
                for (int rb = tid >> 3; rb < nrows; rb += 2 * (kBlock / 8)) {
                    for (int x0 = sub; x0 < rc.ex; x0 += 16) {
                        float val[2][2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            // (addresses clamped instead of loads predicated: a predicated load lands in a basic block
                            // of its own and is waited for right there)
                            const int r = min(rb + u * (kBlock / 8), nrows - 1);
                            const int zrow = (int)(((float)r + 0.5f) * inv_by), yr = r - zrow * by;
                            const int zs = mirror_i32(rc.b0z + zrow, zn->in_len[0]);
                            const int ys = mirror_i32(rc.b0y + yr, zn->in_len[1]);
                            const float* rowp = src + (zs * vol_sz + ys * vol_sy);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int xi = min(x0 + 8 * j, rc.ex - 1);
                                const int xs = mirror_i32(rc.b0x + xi, zn->in_len[2]);
                                val[u][j] = rowp[xs];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int r = rb + u * (kBlock / 8);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int xi = x0 + 8 * j;
                                if (r < nrows && xi < rc.ex) {
                                    box0[r * pitch + xi] = val[u][j];
                                    if (xi > 0)
                                        box1[r * pitch + xi - 1] = val[u][j];
                                }
                            }
                        }
                    }
                }
            }
        };
        // ---- general coordinates of the lane's two voxels of a tile, constant and valid flags -------------------
        int bad = 0;
        auto tile_coords = [&](int t, int mask, const ZRecU& rc, ZVox& vs) {
            const int ez = rc.ez - NT, ey = rc.ey - NT, ex = rc.ex - NTX;
            const bool staged = (rc.flags & kZStaged) != 0;
            vs.flg = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!(mask & (1 << i)))
                    continue;               // (a half of a split tile: the other voxel belongs to the other half)
                const int oz = (sp.tz0 + t) * kT + wave + 4 * i;
                double zw[4];
                k1z_slice(zt, rcol, min(oz, zn->out_len[0] - 1), tp, zw);
                double d[3];
                k1z_disp(tp, zw, d);
                const int b[3] = {oz + zn->off[0], oy + zn->off[1], ox + zn->off[2]};
                double P[3] = {0.0, 0.0, 0.0};
                if (AFFINE) {
#pragma unroll
                    for (int h = 0; h < 3; ++h)
                        P[h] = fma(zn->aff[h * 4 + 0], (double)oz, Pyx[h]);
                }
                int start[3];
                const bool cst = k1z_coords<ORDER, AFFINE>(zn, d, b, P, start, vs.frac[i]);
                const bool valid = oz < zn->out_len[0] && oy < zn->out_len[1] && ox < zn->out_len[2];
                const int rz = start[0] - rc.b0z, ry = start[1] - rc.b0y, rx = start[2] - rc.b0x;
                // (a general tile without a box -- every sample maps to the constant: a voxel that does not is redone)
                if (valid && !cst)
                    bad |= staged ? (rz | ry | rx) | ((ez - rz) | (ey - ry) | (ex - rx)) : -1;
                vs.flg |= (valid && !cst && staged ? 1 << i : 0) | (valid && (cst || staged) ? 4 << i : 0);
                const int off = __mul24(rz, rc.plane) + (__mul24(ry, rc.pitch) + rx);
                vs.addr[i] = __mul24(off & 1, odd_shift) + off * 4;
            }
        };
        // ---- the general tiles of the strip, software-pipelined like the class-A loop.  A work item is a general tile,
        //      or one z half of a tile whose whole box does not fit LDS but whose halves do: the lane's first voxels
        //      (z + 0 .. 3) with the low half's box, then its second voxels with the high half's ------------------------
        {
            // item = 2 * tile + half (a general tile: half 0 only, both voxels)
            auto next_item = [&](int it) {
                int t = it >> 1;
                if (it >= 0 && (it & 1) == 0 && ((sinfo >> (4 * t)) & 7) == 7)
                    return it + 1;
                for (++t; t < ntile; ++t) {
                    const int c = (sinfo >> (4 * t)) & 7;
                    if ((c & 3) == 2 || c == 7)
                        break;
                }
                return 2 * t;
            };
            auto item_rec = [&](int it) {
                const int t = it >> 1;
                const bool split = ((sinfo >> (4 * t)) & 7) == 7;
                return zrec_load(split ? rech0 + ((size_t)t * tile_step * 2 + (it & 1)) : rec0 + (size_t)t * tile_step);
            };
            auto item_mask = [&](int it) { return ((sinfo >> (4 * (it >> 1))) & 7) == 7 ? 1 << (it & 1) : 3; };
            ZVox cur, nxt;
            int it = next_item(-2);
            ZRecU rc = item_rec(it < 2 * ntile ? it : 0), rn = rc;
            if (it < 2 * ntile)
                tile_coords(it >> 1, item_mask(it), rc, cur);
            while (it < 2 * ntile) {
                stage(rc, vol + (STEPS ? steps[0] : 0));
                const int in = next_item(it);
                if (in < 2 * ntile) {
                    rn = item_rec(in);
                    tile_coords(in >> 1, item_mask(in), rn, nxt);
                }
                const bool staged = (rc.flags & kZStaged) != 0;
                const long long ozoff = (long long)((sp.tz0 + (it >> 1)) * kT + wave) * img_sz + obase;
                for (int ss = 0; ss < nsteps; ++ss) {
                    if (STEPS && ss > 0)
                        stage(rc, vol + steps[2 * ss]);
                    if (staged)
                        zdma_barrier();
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (!(cur.flg & (4 << i)))
                            continue;
                        float val = zn->cval;
                        if (cur.flg & (1 << i))
                            val = k1z_voxel<ORDER>(smem, cur.addr[i], cur.frac[i], rc.pitch, rc.plane);
                        zstore_out<OUT16>(img, (STEPS ? steps[2 * ss + 1] : 0) + ozoff + (long long)(4 * i) * img_sz, val, io16);
                    }
                    if (staged)
                        zlds_barrier();
                }
                cur = nxt;
                rc = rn;
                it = in;
            }
        }
        if (__any(bad < 0) && lane == 0) {
            // the fix-up kernel redoes the voxels of this strip whose window is not inside the box
            if (atomicOr(a.missed + sp.id, 1) == 0) {
                const int xcd = vblock & 7;
                a.list_f[(size_t)atomicAdd(a.ctl + zctl(a.parity, 1, xcd), 1) * 8 + xcd] = sp.id;
            }
        }
    }
}

// ONE launch for both: rows of 8 consecutive workgroups (one per XCD) take the role of the fast kernel or -- every
// `every`-th row, until there are `ngen` of them -- of the general kernel, so that the persistent general workgroups start
// with the first class-A strips and run beside them.  (As two launches on two streams the fork and the join cost the
// caller's stream ~6 us of idle each, and at the stream priorities on offer the general kernel either took the class-A
// kernel's slots or was starved until it had finished.)
template <int ORDER, bool AFFINE, bool OUT16, bool STEPS>
__global__ __launch_bounds__(kBlock, 4) void k1z_tile_kernel(const ZFast a, czgen_p zn, const int ngen, const int every)
{
    const int row = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
    const int gen_rows = ngen >> 3;
    // rows every - 1, 2 every - 1, ... are general rows while they last
    const int gens_before = min((row + 1) / every, gen_rows);         // general rows among rows 0 .. row (inclusive)
    const bool is_gen = gens_before > 0 && (row + 1) % every == 0 && (row + 1) / every <= gen_rows;
    if (is_gen)
        k1z_gen_body<ORDER, AFFINE, OUT16, STEPS>(a, zn, (gens_before - 1) * 8 + xcd, ngen);
    else
        k1z_fast_body<ORDER, AFFINE, OUT16, STEPS>(a, (row - gens_before) * 8 + xcd);
}

// What the tile kernels could not serve, straight from global memory: every voxel of a tile whose box does not fit LDS,
// and -- in strips where a tile kernel raised the flag -- the voxels whose window is not inside their tile's sampled
// box.  Persistent over list F.  General coordinates from R, taps mirror-mapped per axis (deform.c:791-813),
// accumulation x, y, z as chains of fused multiply-adds from zero: the bits of the tile loops.  Per wave, no barriers,
// no LDS.
template <int ORDER, bool AFFINE, bool OUT16, bool STEPS>
__global__ __launch_bounds__(kBlock) void k1z_fix_kernel(const ZFast a, czgen_p zn)
{
    constexpr int NT = ORDER + 1;
    constexpr int kPadX = NT & 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = zuni(tid >> 6);
    const int io16 = OUT16 ? a.io16 : 0;
    const int xcd = (int)(blockIdx.x & 7), nper = (int)(gridDim.x >> 3);
    const int nwork = ((cint_p)(const void*)a.ctl)[zctl(a.parity, 1, xcd)];      // (this XCD's share of list F)
    if ((int)(blockIdx.x >> 3) >= nwork)
        return;
    const int vol_sz = a.vol_sz, vol_sy = a.vol_sy, img_sz = a.img_sz;
    const size_t tile_step = (size_t)a.tiles_y * a.tiles_x;
    cdbl_p zt = (cdbl_p)(const void*)a.zt;
    cll_p steps = (cll_p)(const void*)a.steps;
    const int nsteps = STEPS ? a.nsteps : 1;
    const int yy = lane >> 3, xx = lane & 7;

    for (int work = blockIdx.x >> 3; work < nwork; work += nper) {
        ZStrip sp;
        {
            int sid = ((cint_p)(const void*)a.list_f)[(size_t)work * 8 + xcd];
            sp.id = sid;
            sp.sample = sid / a.nstrips;
            sid -= sp.sample * a.nstrips;
            sp.tx = sid % a.tiles_x;
            sid /= a.tiles_x;
            sp.ty = sid % a.tiles_y;
            sp.tz0 = (sid / a.tiles_y) * a.strip_tiles;
            sp.ntile = min(a.strip_tiles, a.tiles_z - sp.tz0);
        }
        const int ntile = sp.ntile;
        crec_p rec0 = (crec_p)(const void*)a.recs + ((size_t)sp.sample * a.ntiles + ((size_t)sp.tz0 * a.tiles_y + sp.ty) * a.tiles_x + sp.tx);
        crec_p rech0 = (crec_p)(const void*)a.recs_half + ((size_t)sp.sample * a.ntiles + ((size_t)sp.tz0 * a.tiles_y + sp.ty) * a.tiles_x + sp.tx) * 2;
        const int sinfo = ((cint_p)(const void*)a.sinfo)[sp.id];
        const bool missed = (((cint_p)(const void*)a.missed)[sp.id] & 1) != 0;
        if (!STEPS && false)
            (void)steps;
        const int oy = sp.ty * kT + yy, ox = sp.tx * kT + xx;
        const int obase = oy * a.img_sy + ox;
        const float* __restrict__ vol = a.vol + sp.sample * a.vol_bstride;
        float* img = a.img + sp.sample * a.img_bstride;
        const char* rcol = reinterpret_cast<const char*>(a.r + sp.sample * a.r_bstride) +
                           ((size_t)min(oy, a.out_y - 1) * a.out_x + min(ox, a.out_x - 1)) * (size_t)a.rcol_bytes;
        double Pyx[3];
#pragma unroll
        for (int h = 0; h < 3; ++h)
            Pyx[h] = AFFINE ? fma(zn->aff[h * 4 + 2], (double)ox, fma(zn->aff[h * 4 + 1], (double)oy, zn->aff[h * 4 + 3] + zn->offd[h])) : 0.0;
        ZTaps tp;
        tp.key[0] = tp.key[1] = tp.key[2] = tp.key[3] = -1;
        {
            if (missed && lane == 0)
                ZSTAT(0, 1);
            // (blockIdx.y = 2 * tile + voxel: a strip's fix-up is dealt to 2 * strip_tiles workgroups -- one workgroup
            // walking a whole tile that does not fit, 128 dependent gathers per lane, took ~90 us, and there are few of them)
            for (int ti = (int)(blockIdx.y >> 1); ti < ntile; ti += ntile) {
                const int c4 = (sinfo >> (4 * ti)) & 7;
                const int cls = c4 & 3;
                const bool split = c4 == 7;
                const bool whole = cls == 3 && !split;
                if (!whole && !missed)
                    continue;
#pragma unroll 1
                for (int i = (int)(blockIdx.y & 1); i < 2; i += 2) {
                    // (a split tile's voxel i worked with the box of half i)
                    const ZRecU rc = zrec_load(split ? rech0 + ((size_t)ti * tile_step * 2 + i) : rec0 + (size_t)ti * tile_step);
                    const int oz = (sp.tz0 + ti) * kT + wave + 4 * i;
                    if (oz >= zn->out_len[0])
                        continue;                   // (uniform)
                    double zw[4];
                    k1z_slice(zt, rcol, oz, tp, zw);
                    if (oy >= zn->out_len[1] || ox >= zn->out_len[2])
                        continue;
                    double d[3];
                    k1z_disp(tp, zw, d);
                    const int b[3] = {oz + zn->off[0], oy + zn->off[1], ox + zn->off[2]};
                    double P[3] = {0.0, 0.0, 0.0};
                    if (AFFINE) {
#pragma unroll
                        for (int h = 0; h < 3; ++h)
                            P[h] = fma(zn->aff[h * 4 + 0], (double)oz, Pyx[h]);
                    }
                    int st[3], raw[3];
                    float fr[3];
                    const bool cst = k1z_coords<ORDER, AFFINE>(zn, d, b, P, st, fr, raw);
                    if (!whole) {
                        // A class-A tile's voxels worked with the RAW window start (no range test), a general tile's
                        // with the mapped one; the loop served the voxel iff that window lay inside the box (a constant
                        // voxel of a general tile needs no window).
                        const bool gen = cls != 1;
                        if (gen && cst)
                            continue;
                        const int rz = (gen ? st[0] : raw[0]) - rc.b0z, ry = (gen ? st[1] : raw[1]) - rc.b0y,
                                  rx = (gen ? st[2] : raw[2]) - rc.b0x;
                        const bool inside = rz >= 0 && rz + NT <= rc.ez && ry >= 0 && ry + NT <= rc.ey && rx >= 0 &&
                                            rx + NT + kPadX <= rc.ex;
                        if (inside)
                            continue;
                    }
                    int tap[3][NT];
                    float w[3][NT];
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        weights_from_frac<float, ORDER>(fr[h], w[h]);
                        const int stride = h == 0 ? vol_sz : (h == 1 ? vol_sy : 1);
#pragma unroll
                        for (int l = 0; l < NT; ++l)
                            tap[h][l] = mirror_i32(st[h] + l, zn->in_len[h]) * stride;
                    }
                    const long long ooff = (long long)oz * img_sz + obase;
                    ZSTAT(whole ? 2 : 1, 1);
                    for (int ss = 0; ss < nsteps; ++ss) {
                        float val = zn->cval;
                        if (!cst) {
                            const float* src = vol + (STEPS ? steps[2 * ss] : 0);
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
                        zstore_out<OUT16>(img, (STEPS ? steps[2 * ss + 1] : 0) + ooff, val, io16);
                    }
                }
            }
        }
    }
}

template <int ORDER, bool AFFINE, bool OUT16, bool STEPS>
hipError_t launch_k1z_kernels(const ZFast& zf, const void* znp, unsigned ngen, unsigned nfix, size_t lds, hipStream_t stream, SideLane*)
{
    const unsigned nfast = k1z_grid(zf.total_strips, zf.deal);
    // general rows interleaved with the fast rows in proportion, all of them within the first fast rows' reach
    const unsigned fast_rows = nfast >> 3, gen_rows = ngen >> 3;
    unsigned every = gen_rows ? fast_rows / gen_rows + 1 : 2;
    if (every < 2)
        every = 2;
    hipLaunchKernelGGL((k1z_tile_kernel<ORDER, AFFINE, OUT16, STEPS>), dim3(nfast + ngen), dim3(kBlock), lds, stream, zf, (czgen_p)znp,
                       (int)ngen, (int)every);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL((k1z_fix_kernel<ORDER, AFFINE, OUT16, STEPS>), dim3(nfix, 2 * zf.strip_tiles), dim3(kBlock), 0, stream, zf, (czgen_p)znp);
    return hipGetLastError();
}
template <int ORDER, bool AFFINE, bool OUT16>
hipError_t launch_k1z_variant(const ZFast& zf, const void* zn, unsigned ngen, unsigned nfix, size_t lds, hipStream_t stream, SideLane* side,
                              bool steps)
{
    return steps ? launch_k1z_kernels<ORDER, AFFINE, OUT16, true>(zf, zn, ngen, nfix, lds, stream, side)
                 : launch_k1z_kernels<ORDER, AFFINE, OUT16, false>(zf, zn, ngen, nfix, lds, stream, side);
}
template <int ORDER>
hipError_t launch_k1z_order(const HotGeom& hg, const ZFast& zf, const void* zn, unsigned ngen, unsigned nfix, size_t lds, hipStream_t stream,
                            SideLane* side)
{
    const bool steps = hg.nstep != 0;
    if (hg.io16)
        return hg.has_affine ? launch_k1z_variant<ORDER, true, true>(zf, zn, ngen, nfix, lds, stream, side, steps)
                             : launch_k1z_variant<ORDER, false, true>(zf, zn, ngen, nfix, lds, stream, side, steps);
    return hg.has_affine ? launch_k1z_variant<ORDER, true, false>(zf, zn, ngen, nfix, lds, stream, side, steps)
                         : launch_k1z_variant<ORDER, false, false>(zf, zn, ngen, nfix, lds, stream, side, steps);
}

}  // namespace

// LDS of both kernels: the box pair.  Five workgroups per
// CU -> 31 KiB each (3904 floats per copy: a 13 x 13 x 16 box of the benchmark field takes 2704); large boxes (52 KiB,
// three per CU) for strongly deformed volumes, chosen by the spill feedback
size_t k1z_lds_bytes(int* box_cap, bool large)
{
    // (large: 52 KiB, three workgroups per CU -- against 40 KiB / four per CU: forward call at sigma 10 / 12.5 / 15 / 20
    // 248 / 288 / 343 / 609 -> 239 / 292 / 331 / 457 us, profiles/r06_k1_box_sweep.txt)
    size_t budget = large ? 52 * 1024 : 31 * 1024;
    if (const char* kb = ed_env("EDHIP_ZKB"))
        budget = (size_t)atoi(kb) * 1024;
    size_t cap = budget / 8;
    cap = ((cap - 56) / 64) * 64 + 56;        // cap = 56 (mod 64): the copies sit on disjoint banks
    *box_cap = (int)cap;
    return 2 * 4 * cap;
}

size_t k1z_geo_lds_bytes(const GridGeom& g, const HotGeom& hg)
{
    const size_t ngrid = 3 * (size_t)g.ncp[0] * (size_t)g.ncp[1] * (size_t)g.ncp[2];
    return 8 * ngrid + 8 * 64 * 3 * (size_t)g.ncp[0] + 16 * sizeof(AxTab) + 3 * kGeoWaves * 8 + (sizeof(AxTab) * 4 + 4 + 256) * (size_t)hg.tiles[0] + 16;
}

size_t k1z_r_bytes(const GridGeom& g) { return 8 * (size_t)g.out_len[1] * (size_t)g.out_len[2] * 4 * (size_t)g.ncp[0]; }

bool k1z_supported(const GridGeom& g)
{
    const size_t ngrid = 3 * (size_t)g.ncp[0] * (size_t)g.ncp[1] * (size_t)g.ncp[2];
    const size_t tiles_z = (size_t)((g.out_len[0] + kT - 1) / kT);
    return ngrid <= 4096 && g.ncp[0] <= 16 &&
           8 * ngrid + 8 * 64 * 3 * (size_t)g.ncp[0] + 16 * sizeof(AxTab) + 3 * kGeoWaves * 8 + (sizeof(AxTab) * 4 + 4 + 256) * tiles_z + 16 <= 60 * 1024;
}

hipError_t launch_k1z_geo(const GridGeom& g, const HotGeom& hg, const ZGeom& zg, const GridPrefilter& gp, int nbatch,
                          hipStream_t stream)
{
    unsigned fill = 0;
    if (zg.zero_ptr && zg.zero_bytes > 0) {
        const long long want = (zg.zero_bytes + 65535) / 65536;       // 64 KiB per workgroup and round
        fill = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    }
    hipLaunchKernelGGL(k1z_geo_kernel, dim3((unsigned)(hg.tiles[1] * hg.tiles[2]) + fill, (unsigned)nbatch), dim3(kGeoBlock),
                       k1z_geo_lds_bytes(g, hg), stream, g, hg, zg, gp);
    if (ed_env("EDHIP_GEO_TWICE"))        // (profiling build: the launch again, warm -- what of its 34 us is a cold instruction cache?)
        hipLaunchKernelGGL(k1z_geo_kernel, dim3((unsigned)(hg.tiles[1] * hg.tiles[2]) + fill, (unsigned)nbatch), dim3(kGeoBlock),
                           k1z_geo_lds_bytes(g, hg), stream, g, hg, zg, gp);
    return hipGetLastError();
}

hipError_t launch_k1z(const HotGeom& hg, const ZGeom& zg, int order, size_t lds, hipStream_t stream, SideLane* side)
{
    ZFast zf;
    memset(&zf, 0, sizeof(zf));
    zf.vol = hg.vol_r;
    zf.img = hg.img_w;
    zf.r = zg.r;
    zf.zt = zg.zt;
    zf.recs = zg.recs;
    zf.recs_half = zg.recs_half;
    zf.missed = zg.missed;
    zf.hint = hg.hint;
    zf.sinfo = zg.sinfo;
    zf.list_g = zg.list_g;
    zf.list_f = zg.list_f;
    zf.ctl = zg.ctl;
    zf.parity = zg.parity;
    zf.steps = zg.steps;
    zf.vol_bstride = hg.vol_bstride;
    zf.img_bstride = hg.img_bstride;
    zf.r_bstride = zg.r_bstride;
    zf.vol_sy = hg.vol_sy;
    zf.vol_sz = hg.vol_sz;
    zf.img_sy = hg.img_sy;
    zf.img_sz = hg.img_sz;
    zf.box_cap = hg.box_cap;
    zf.tiles_z = hg.tiles[0];
    zf.tiles_y = hg.tiles[1];
    zf.tiles_x = hg.tiles[2];
    zf.strip_tiles = zg.strip_tiles;
    zf.nstrips = zg.nstrips;
    zf.total_strips = zg.total_strips;
    zf.ntiles = hg.ntiles;
    zf.rcol_bytes = 32 * zg.ncpz;
    zf.out_y = hg.out_len[1];
    zf.out_x = hg.out_len[2];
    const int H = order / 2;
    zf.kz0 = (hg.has_affine ? 0 : hg.off[0]) - H;
    zf.ky0 = (hg.has_affine ? 0 : hg.off[1]) - H;
    zf.kx0 = (hg.has_affine ? 0 : hg.off[2]) - H;
    zf.io16 = hg.io16;
    zf.nsteps = (int)hg.nsteps;
    zf.dbg = hg.dbg;
    // chunks of a few rows of strips; small launches: smaller chunks, so that every XCD gets some
    zf.deal = 4 * hg.tiles[2];
    while (zf.deal > 1 && zg.total_strips < 16 * zf.deal)
        zf.deal >>= 1;
    if (const char* dl = ed_env("EDHIP_ZDEAL"))
        zf.deal = atoi(dl) >= 1 ? atoi(dl) : zf.deal;
    for (int h = 0; h < 3; ++h) {
        for (int k = 0; k < 3; ++k)
            zf.aff[h * 4 + k] = hg.affine[h * 4 + k];
        zf.aff[h * 4 + 3] = hg.affine[h * 4 + 3] + (double)hg.off[h];
    }
    const void* zn = zg.zgen;
    // persistent grids: the general tiles' list is worked off by up to 5 workgroups per CU; the fix-up list is empty on
    // a mild field (a launch of idle workgroups: ~2 us)
    unsigned ngen = (unsigned)(zg.total_strips < 1280 ? ((zg.total_strips + 7) / 8) * 8 : 1280);
    if (const char* ng = ed_env("EDHIP_ZNGEN"))
        ngen = (unsigned)((atoi(ng) + 7) / 8 * 8);
    const unsigned nfix = (unsigned)(zg.total_strips < 512 ? ((zg.total_strips + 7) / 8) * 8 : 512);
    switch (order) {
    case 1: return launch_k1z_order<1>(hg, zf, zn, ngen, nfix, lds, stream, side);
    case 2: return launch_k1z_order<2>(hg, zf, zn, ngen, nfix, lds, stream, side);
    case 3: return launch_k1z_order<3>(hg, zf, zn, ngen, nfix, lds, stream, side);
    default: return hipErrorNotSupported;
    }
}

#ifdef EDHIP_K1_STATS
// read (and clear) the counters
extern "C" int edhip_debug_k1z_stats(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_k1z_stats), sizeof(z)) != hipSuccess)
        return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k1z_stats), z, sizeof(z)) != hipSuccess;
}
#endif

}  // namespace tile
}  // namespace ed
