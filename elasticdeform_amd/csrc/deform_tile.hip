// deform_tile.hip -- K1 forward, LDS-tiled: the hot kernel of the benchmark workload
// (3 deformed axes, float32 / float64 volumes, spline order 2-5).
//
// Per-voxel pipeline of DeformGrid's forward branch (deform.c:649-924), organised per OUTPUT TILE:
//
//   tile     8 x 8 x 8 output voxels per 256-thread workgroup (4 waves x 2 z-slices each; a lane
//            owns one (y, x) column of the tile), cubic so that the source bounding box stays
//            small under shear (long-x tiles overfetch 6x, cubes 3.5x)
//   phase A  displacement at every voxel (deform.c:650-758), fp64, evaluated separably through
//            LDS by the whole workgroup: the control grid D is parked in LDS, contracted over z
//            for the tile's 8 slices (P), then over y for the 64 rows (Q); a voxel is left with 4
//            x-taps per component (12 fp64 FMAs instead of the reference's 192 multiply-adds).
//            The per-axis weights / mirror-mapped control indices of the tile's 8+8+8 output
//            indices are the block prologue's LDS table (the reference's `dsplvals`,
//            deform.c:639-647).  Then affine, + offset, boundary map, floor -- all fp64 -- and the
//            fractional offsets are handed to fp32 (float32 volumes) for the basis weights.
//   phase B  bounding box of all tap windows of the tile, in UNMAPPED tap-index space
//            (wave min/max reduce -> one LDS atomic per wave).
//   phase C  the source box is staged from HBM/L2 into LDS once (it overlays D/P/Q, which are dead
//            by then), rows coalesced along the fastest axis; every box index goes through the
//            mirror map here, which is exactly what the reference does with the taps of a window
//            that sticks out (deform.c:791-813) -- so the gather needs no edge handling at all.
//            float32: a second copy shifted by one element makes every x-run of taps aligned
//            ds_read_b64 pairs instead of single ds_read_b32.
//   phase D  (order+1)^3 tap gather from LDS, accumulated separably (x, y, z) in the data's width.
//   spill    a tile whose box exceeds the LDS budget (strong folding, 'wrap' seams) is appended to
//            a worklist and finished by deform_tile3_spill_kernel straight from global memory with
//            per-tap mirror mapping; the hot kernel carries no fallback code.
//
// HBM traffic: each source voxel is fetched ~3.5x per launch but from L2 / Infinity Cache
// (neighbouring tiles overlap; tiles are dealt to the 8 XCDs in contiguous chunks so that the
// overlap stays inside one L2); algorithmic bytes are 4 read + 4 written per voxel (float32).
#include "ed_device.h"
#include "ed_params.h"

namespace ed {

namespace {

constexpr int kT = 8;                 // tile edge
constexpr int kBlock = 256;
constexpr int kTabBytes = 3 * kT * 48;             // 1152
constexpr int kRedInts = 8;
constexpr int kHeadBytes = kTabBytes + kRedInts * 4;   // 1184, multiple of 16

struct AxTab {
    double w[4];
    int idx[4];
};
static_assert(sizeof(AxTab) == 48, "AxTab layout");
static_assert(kHeadBytes % 16 == 0, "LDS carve alignment");

__device__ __forceinline__ int mirror_i32(int idx, int len)
{
    if (len <= 1)
        return 0;
    const int period = 2 * len - 2;
    if (idx < 0) {
        idx = period * (-idx / period) + idx;
        idx = idx <= 1 - len ? idx + period : -idx;
    } else if (idx >= len) {
        idx -= period * (idx / period);
        if (idx >= len)
            idx = period - idx;
    }
    return idx;
}

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = min(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = max(v, __shfl_xor(v, m, 64));
    return v;
}

// B-spline basis weights from the fractional offset, in the data's own width.  Same closed forms
// as deform.c:160-268 (last weight = 1 - sum of the others) with the divisions folded into
// constants.  x is c - floor(c) (odd orders) or c - floor(c + 0.5) (even orders).
template <typename T, int ORDER>
__device__ __forceinline__ void weights_from_frac(T x, T* w)
{
    const T z = (T)1 - x;
    if (ORDER == 0) {
        w[0] = (T)1;
        return;
    }
    if (ORDER == 1) {
        w[0] = z;
    } else if (ORDER == 2) {
        w[1] = (T)0.75 - x * x;
        const T y = (T)0.5 - x;
        w[0] = (T)0.5 * y * y;
    } else if (ORDER == 3) {
        w[1] = (x * x * (x - (T)2) * (T)3 + (T)4) * (T)(1.0 / 6.0);
        w[2] = (z * z * (z - (T)2) * (T)3 + (T)4) * (T)(1.0 / 6.0);
        w[0] = z * z * z * (T)(1.0 / 6.0);
    } else if (ORDER == 4) {
        T t = x * x;
        w[2] = t * (t * (T)0.25 - (T)0.625) + (T)(115.0 / 192.0);
        T y = (T)1 + x;
        w[1] = y * (y * (y * ((T)5 - y) * (T)(1.0 / 6.0) - (T)1.25) + (T)(5.0 / 24.0)) +
               (T)(55.0 / 96.0);
        w[3] = z * (z * (z * ((T)5 - z) * (T)(1.0 / 6.0) - (T)1.25) + (T)(5.0 / 24.0)) +
               (T)(55.0 / 96.0);
        y = (T)0.5 - x;
        t = y * y;
        w[0] = t * t * (T)(1.0 / 24.0);
    } else {
        T t = x * x;
        w[2] = t * (t * ((T)0.25 - x * (T)(1.0 / 12.0)) - (T)0.5) + (T)0.55;
        t = z * z;
        w[3] = t * (t * ((T)0.25 - z * (T)(1.0 / 12.0)) - (T)0.5) + (T)0.55;
        T y = x + (T)1;
        w[1] = y * (y * (y * (y * (y * (T)(1.0 / 24.0) - (T)0.375) + (T)1.25) - (T)1.75) +
                    (T)0.625) + (T)0.425;
        const T zz = z + (T)1;
        w[4] = zz * (zz * (zz * (zz * (zz * (T)(1.0 / 24.0) - (T)0.375) + (T)1.25) - (T)1.75) +
                     (T)0.625) + (T)0.425;
        y = (T)1 - x;
        t = y * y;
        w[0] = y * t * t * (T)(1.0 / 120.0);
    }
    T last = (T)1;
#pragma unroll
    for (int i = 0; i < ORDER; ++i)
        last -= w[i];
    w[ORDER] = last;
}

struct TileGeom {
    int in_len[3];        // I_k (the tile kernels require extents < 2^30)
    int out_len[3];
    int tiles[3];         // number of tiles per axis
    int ntiles;
    int64_t in_stride[3];    // element strides
    int64_t out_stride[3];
    int box_cap;          // elements per LDS copy
    int overlay_bytes;    // size of the D/P/Q | box overlay region
    int* spill;           // [0] = count, [1..] = tile ids that did not fit in LDS
};

__device__ __forceinline__ void tile_origin(const TileGeom& tg, int t, int* o0)
{
    const int tx = t % tg.tiles[2];
    t /= tg.tiles[2];
    const int ty = t % tg.tiles[1];
    const int tz = t / tg.tiles[1];
    o0[0] = tz * kT;
    o0[1] = ty * kT;
    o0[2] = tx * kT;
}

template <typename T, int ORDER, bool PAIR>
__global__ __launch_bounds__(kBlock, 4) void deform_tile3_fwd_kernel(const GridGeom g,
                                                                     const IOView v,
                                                                     const TileGeom tg)
{
    constexpr int NT = ORDER + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AxTab* tab = reinterpret_cast<AxTab*>(smem);                      // [3][8]
    int* sred = reinterpret_cast<int*>(smem + kTabBytes);              // lo[3], hi[3]
    char* overlay = smem + kHeadBytes;
    // phase A view of the overlay
    const int ncpz = (int)g.ncp[0], ncpy = (int)g.ncp[1], ncpx = (int)g.ncp[2];
    const int nyx = ncpy * ncpx;
    double* sD = reinterpret_cast<double*>(overlay);         // [3][ncpz][nyx]
    double* sP = sD + 3 * ncpz * nyx;                         // [8][3][nyx]
    double* sQ = sP + kT * 3 * nyx;                           // [8][8][3][ncpx]
    // phase C/D view
    T* box0 = reinterpret_cast<T*>(overlay);
    T* box1 = box0 + tg.box_cap + 8;      // +8 elements: the two copies sit on disjoint LDS banks

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int yy = lane >> 3, xx = lane & 7;

    // tiles are dealt to the XCDs in contiguous chunks (block b runs on XCD b % 8): neighbouring
    // tiles, whose source boxes overlap, share an L2
    int tile;
    {
        const int b = blockIdx.x;
        const int per = (tg.ntiles + 7) >> 3;
        tile = (b & 7) * per + (b >> 3);
        if (tile >= tg.ntiles)
            return;
    }
    int o0[3];
    tile_origin(tg, tile, o0);

    // ---- block prologue: per-axis displacement weights / control indices (deform.c:639-690),
    //      control grid -> LDS as doubles ----------------------------------------------------------
    if (tid < 3 * kT) {
        const int a = tid >> 3, i = tid & 7;
        int oi = o0[a] + i;
        if (oi >= tg.out_len[a])
            oi = tg.out_len[a] - 1;
        const double cp = control_coordinate(g.ncp[a], (int64_t)oi + g.off[a], g.in_len[a]);
        const int64_t start = window_start(cp, 3);
        const bool edge = start < 0 || start + 3 >= g.ncp[a];
        double w[4];
        spline_weights(cp, 3, w);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            tab[tid].w[l] = w[l];
            tab[tid].idx[l] = (int)(edge ? mirror_index(start + l, g.ncp[a]) : start + l);
        }
    }
    if (tid < 3) {
        sred[tid] = 0x7fffffff;
        sred[3 + tid] = (int)0x80000000;
    }
    for (int e = tid; e < 3 * ncpz * nyx; e += kBlock) {
        int r = e;
        const int j2 = r % ncpx;
        r /= ncpx;
        const int j1 = r % ncpy;
        r /= ncpy;
        const int j0 = r % ncpz;
        const int h = r / ncpz;
        sD[e] = load_as_double(g.disp + g.disp_stride[0] * h + g.disp_stride[1] * j0 +
                                   g.disp_stride[2] * j1 + g.disp_stride[3] * j2,
                               g.disp_dtype);
    }
    __syncthreads();

    // ---- phase A.1: P[zi][h][j] = sum_l wz[zi][l] * D[h][iz[zi][l]][j] ----------------------------
    for (int e = tid; e < kT * 3 * nyx; e += kBlock) {
        const int zi = e / (3 * nyx), r = e - zi * 3 * nyx;
        const int h = r / nyx, j = r - h * nyx;
        const AxTab& tz_ = tab[zi];
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l)
            acc += tz_.w[l] * sD[(h * ncpz + tz_.idx[l]) * nyx + j];
        sP[e] = acc;
    }
    __syncthreads();
    // ---- phase A.2: Q[zi][y][h][j2] = sum_l wy[y][l] * P[zi][h][iy[y][l]][j2] -----------------------
    for (int q = tid; q < kT * kT * 3 * ncpx; q += kBlock) {
        int r = q;
        const int j2 = r % ncpx;
        r /= ncpx;
        const int h = r % 3;
        r /= 3;
        const int y = r & 7, zi = r >> 3;
        const AxTab& ty_ = tab[kT + y];
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l)
            acc += ty_.w[l] * sP[(zi * 3 + h) * nyx + ty_.idx[l] * ncpx + j2];
        sQ[q] = acc;
    }
    __syncthreads();

    // ---- phase A.3: coordinates of this thread's two voxels --------------------------------------
    int start[2][3];
    T frac[2][3];
    bool valid[2], constant[2];
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    {
        const AxTab& tx_ = tab[2 * kT + xx];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int zi = wave + 4 * i;
            const int o[3] = {o0[0] + zi, o0[1] + yy, o0[2] + xx};
            valid[i] = o[0] < tg.out_len[0] && o[1] < tg.out_len[1] && o[2] < tg.out_len[2];
            bool cst = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const double* qrow = sQ + ((zi * kT + yy) * 3 + h) * ncpx;
                double d = 0.0;
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    d += tx_.w[l] * qrow[tx_.idx[l]];
                double c;
                if (g.has_affine) {
                    c = g.affine[h * 4 + 3];
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                        c += g.affine[h * 4 + l] * (double)o[l];
                } else {
                    c = (double)o[h];
                }
                c = map_coordinate(c + (double)g.off[h] + d, g.in_len[h], v.mode);
                const bool bad = !(c > -1.0);
                cst = cst || bad;
                const double fl = floor((ORDER & 1) ? c : c + 0.5);
                start[i][h] = bad ? 0 : (int)fl - ORDER / 2;
                frac[i][h] = (T)(c - fl);
            }
            constant[i] = cst;
            if (valid[i] && !cst) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    lo[h] = min(lo[h], start[i][h]);
                    hi[h] = max(hi[h], start[i][h] + ORDER);
                }
            }
        }
    }

    // ---- phase B: bounding box of the tile's tap windows ------------------------------------------
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const int l = wave_min(lo[h]);
        const int u = wave_max(hi[h]);
        if (lane == 0) {
            atomicMin(&sred[h], l);
            atomicMax(&sred[3 + h], u);
        }
    }
    __syncthreads();      // also: every read of Q is done, the overlay may become the box
    const int b0[3] = {sred[0], sred[1], sred[2]};
    const int ext[3] = {sred[3] - sred[0] + 1, sred[4] - sred[1] + 1, sred[5] - sred[2] + 1};
    const bool any = sred[3] >= sred[0];
    // row pitch: PAIR (ds_read_b64): 16 * odd puts 4 consecutive rows on 4 disjoint bank groups;
    // b32 / f64 reads: 8 or 24 (mod 32)
    int pitch;
    if (PAIR)
        pitch = ext[2] <= 16 ? 16 : (ext[2] <= 48 ? 48 : 0);
    else
        pitch = ext[2] <= 8 ? 8 : (ext[2] <= 24 ? 24 : (ext[2] <= 56 ? 56 : 0));
    const int by = ext[1];
    const int nrows = ext[0] * by;
    const bool fits = any && pitch > 0 && (int64_t)nrows * pitch <= tg.box_cap;
    if (any && !fits) {
        // hand the whole tile to the spill kernel
        if (tid == 0) {
            const int slot = atomicAdd(&tg.spill[0], 1);
            tg.spill[1 + slot] = tile;
        }
        return;
    }
    const bool x_inside = b0[2] >= 0 && b0[2] + ext[2] <= tg.in_len[2];

    const T* __restrict__ in = reinterpret_cast<const T*>(v.in);
    T* out = reinterpret_cast<T*>(v.out);

    for (int64_t ss = 0; ss < v.nsteps; ++ss) {
        int64_t in_off = 0, out_off = 0;
        {
            int64_t r = ss;
            for (int l = 0; l < v.nstep; ++l) {
                const int64_t q = r / v.step_len[l];
                const int64_t c = r - q * v.step_len[l];
                in_off += v.in_step_stride[l] * c;
                out_off += v.out_step_stride[l] * c;
                r = q;
            }
        }
        const T* src = in + in_off;

        if (any) {
            // ---- phase C: stage the source box (mirror-mapped) into LDS ---------------------------
            if (ss > 0)
                __syncthreads();     // previous step's gathers are done with the box
            const int sub = tid & 7;
            for (int r = tid >> 3; r < nrows; r += kBlock / 8) {
                const int zr = r / by, yr = r - zr * by;
                const int zs = mirror_i32(b0[0] + zr, tg.in_len[0]);
                const int ys = mirror_i32(b0[1] + yr, tg.in_len[1]);
                const T* rowp = src + (int64_t)zs * tg.in_stride[0] + (int64_t)ys * tg.in_stride[1];
                T* d0 = box0 + r * pitch;
                T* d1 = box1 + r * pitch;
                for (int xi = sub; xi < ext[2]; xi += 8) {
                    const int xs = x_inside ? b0[2] + xi : mirror_i32(b0[2] + xi, tg.in_len[2]);
                    const T val = rowp[(int64_t)xs * tg.in_stride[2]];
                    d0[xi] = val;
                    if (PAIR && xi > 0)
                        d1[xi - 1] = val;
                }
            }
            __syncthreads();
        }

        // ---- phase D: gather ------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!valid[i])
                continue;
            T val;
            if (constant[i]) {
                val = (T)v.cval;
            } else {
                T w0[NT], w1[NT], w2[NT];
                weights_from_frac<T, ORDER>(frac[i][0], w0);
                weights_from_frac<T, ORDER>(frac[i][1], w1);
                weights_from_frac<T, ORDER>(frac[i][2], w2);
                const int rz = start[i][0] - b0[0], ry = start[i][1] - b0[1],
                          rx = start[i][2] - b0[2];
                const int rowbase = (rz * by + ry) * pitch;
                T a0 = 0;
                if (PAIR) {
                    // consecutive x-taps as aligned 8-byte reads from the copy whose shift matches
                    // the parity of rx
                    const T* bp = (rx & 1) ? box1 + rowbase + rx - 1 : box0 + rowbase + rx;
#pragma unroll
                    for (int l0 = 0; l0 < NT; ++l0) {
                        T a1 = 0;
#pragma unroll
                        for (int l1 = 0; l1 < NT; ++l1) {
                            const T* rp = bp + (l0 * by + l1) * pitch;
                            T a2 = 0;
#pragma unroll
                            for (int l2 = 0; l2 < NT; l2 += 2) {
                                const float2 pr = *reinterpret_cast<const float2*>(rp + l2);
                                a2 += w2[l2] * pr.x;
                                a2 += w2[l2 + 1] * pr.y;
                            }
                            a1 += w1[l1] * a2;
                        }
                        a0 += w0[l0] * a1;
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
                                a2 += w2[l2] * rp[l2];
                            a1 += w1[l1] * a2;
                        }
                        a0 += w0[l0] * a1;
                    }
                }
                val = a0;
            }
            const int oz = o0[0] + wave + 4 * i, oy = o0[1] + yy, ox = o0[2] + xx;
            out[out_off + (int64_t)oz * tg.out_stride[0] + (int64_t)oy * tg.out_stride[1] +
                (int64_t)ox * tg.out_stride[2]] = val;
        }
    }
}

// Tiles that did not fit in LDS: one thread per voxel, taps straight from global memory with the
// per-tap mirror map of deform.c:791-813; displacement by the direct 64-tap sum (deform.c:693-758).
template <typename T, int ORDER>
__global__ __launch_bounds__(kBlock) void deform_tile3_spill_kernel(const GridGeom g, const IOView v,
                                                                    const TileGeom tg)
{
    constexpr int NT = ORDER + 1;
    const int nspill = tg.spill[0];
    const T* __restrict__ in = reinterpret_cast<const T*>(v.in);
    T* out = reinterpret_cast<T*>(v.out);
    for (int s = blockIdx.x; s < nspill; s += gridDim.x) {
        int o0[3];
        tile_origin(tg, tg.spill[1 + s], o0);
        for (int vox = threadIdx.x; vox < kT * kT * kT; vox += kBlock) {
            const int o[3] = {o0[0] + (vox >> 6), o0[1] + ((vox >> 3) & 7), o0[2] + (vox & 7)};
            if (o[0] >= tg.out_len[0] || o[1] >= tg.out_len[1] || o[2] >= tg.out_len[2])
                continue;
            double dw[3][4];
            int64_t dtap[3][4];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double cp = control_coordinate(g.ncp[k], (int64_t)o[k] + g.off[k], g.in_len[k]);
                const int64_t st = window_start(cp, 3);
                const bool edge = st < 0 || st + 3 >= g.ncp[k];
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    dtap[k][l] = (edge ? mirror_index(st + l, g.ncp[k]) : st + l) * g.disp_stride[k + 1];
                spline_weights(cp, 3, dw[k]);
            }
            int start[3];
            T w[3][NT];
            bool cst = false;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const char* base = g.disp + g.disp_stride[0] * h;
                double d = 0.0;
                for (int t = 0; t < 64; ++t) {
                    const int l0 = t >> 4, l1 = (t >> 2) & 3, l2 = t & 3;
                    d += load_as_double(base + dtap[0][l0] + dtap[1][l1] + dtap[2][l2], g.disp_dtype) *
                         dw[0][l0] * dw[1][l1] * dw[2][l2];
                }
                double c;
                if (g.has_affine) {
                    c = g.affine[h * 4 + 3];
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                        c += g.affine[h * 4 + l] * (double)o[l];
                } else {
                    c = (double)o[h];
                }
                c = map_coordinate(c + (double)g.off[h] + d, g.in_len[h], v.mode);
                const bool bad = !(c > -1.0);
                cst = cst || bad;
                const double fl = floor((ORDER & 1) ? c : c + 0.5);
                start[h] = bad ? 0 : (int)fl - ORDER / 2;
                weights_from_frac<T, ORDER>((T)(c - fl), w[h]);
            }
            const int64_t obase = (int64_t)o[0] * tg.out_stride[0] + (int64_t)o[1] * tg.out_stride[1] +
                                  (int64_t)o[2] * tg.out_stride[2];
            for (int64_t ss = 0; ss < v.nsteps; ++ss) {
                int64_t in_off = 0, out_off = 0, r = ss;
                for (int l = 0; l < v.nstep; ++l) {
                    const int64_t q = r / v.step_len[l];
                    const int64_t c = r - q * v.step_len[l];
                    in_off += v.in_step_stride[l] * c;
                    out_off += v.out_step_stride[l] * c;
                    r = q;
                }
                T val;
                if (cst) {
                    val = (T)v.cval;
                } else {
                    const T* src = in + in_off;
                    T a0 = 0;
                    for (int l0 = 0; l0 < NT; ++l0) {
                        const T* p0 = src + (int64_t)mirror_i32(start[0] + l0, tg.in_len[0]) * tg.in_stride[0];
                        T a1 = 0;
                        for (int l1 = 0; l1 < NT; ++l1) {
                            const T* p1 = p0 + (int64_t)mirror_i32(start[1] + l1, tg.in_len[1]) * tg.in_stride[1];
                            T a2 = 0;
#pragma unroll
                            for (int l2 = 0; l2 < NT; ++l2)
                                a2 += w[2][l2] * p1[(int64_t)mirror_i32(start[2] + l2, tg.in_len[2]) * tg.in_stride[2]];
                            a1 += w[1][l1] * a2;
                        }
                        a0 += w[0][l0] * a1;
                    }
                    val = a0;
                }
                out[out_off + obase] = val;
            }
        }
    }
}

inline size_t dpq_bytes(const GridGeom& g)
{
    const size_t nyx = (size_t)g.ncp[1] * g.ncp[2];
    return 8 * (3 * (size_t)g.ncp[0] * nyx + kT * 3 * nyx + (size_t)kT * kT * 3 * g.ncp[2]);
}

template <bool PAIR, typename T>
constexpr int box_cap()
{
    return PAIR ? 4096 : (sizeof(T) == 4 ? 6144 : 4096);
}
template <bool PAIR, typename T>
constexpr size_t box_bytes()
{
    return (PAIR ? 2 * (size_t)box_cap<PAIR, T>() + 8 : (size_t)box_cap<PAIR, T>()) * sizeof(T);
}

template <typename T, int ORDER, bool PAIR>
hipError_t launch_tile(const GridGeom& g, const IOView& v, hipStream_t stream)
{
    TileGeom tg;
    IOView ve = v;
    int64_t ntiles = 1;
    for (int k = 0; k < 3; ++k) {
        tg.in_len[k] = (int)g.in_len[k];
        tg.out_len[k] = (int)g.out_len[k];
        tg.tiles[k] = (int)((g.out_len[k] + kT - 1) / kT);
        tg.in_stride[k] = v.in_stride[k] / (int64_t)sizeof(T);
        tg.out_stride[k] = v.out_stride[k] / (int64_t)sizeof(T);
        ntiles *= tg.tiles[k];
    }
    for (int l = 0; l < v.nstep; ++l) {
        ve.in_step_stride[l] = v.in_step_stride[l] / (int64_t)sizeof(T);
        ve.out_step_stride[l] = v.out_step_stride[l] / (int64_t)sizeof(T);
    }
    if (ntiles <= 0)
        return hipSuccess;
    if (ntiles > 0x3fffffffLL)
        return hipErrorInvalidValue;
    tg.ntiles = (int)ntiles;
    tg.box_cap = box_cap<PAIR, T>();
    size_t overlay = box_bytes<PAIR, T>();
    if (dpq_bytes(g) > overlay)
        overlay = dpq_bytes(g);
    overlay = (overlay + 15) & ~(size_t)15;
    tg.overlay_bytes = (int)overlay;
    const size_t lds = kHeadBytes + overlay;

    // spill worklist: stream-ordered scratch, counter zeroed on the stream
    void* spill = nullptr;
    hipError_t e = hipMallocAsync(&spill, sizeof(int) * ((size_t)ntiles + 1), stream);
    if (e != hipSuccess)
        return e;
    tg.spill = (int*)spill;
    e = hipMemsetAsync(spill, 0, sizeof(int), stream);
    if (e == hipSuccess) {
        const unsigned nblk = (unsigned)(((ntiles + 7) / 8) * 8);
        hipLaunchKernelGGL((deform_tile3_fwd_kernel<T, ORDER, PAIR>), dim3(nblk), dim3(kBlock), lds,
                           stream, g, ve, tg);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const unsigned nsp = (unsigned)(ntiles < 2048 ? ntiles : 2048);
        hipLaunchKernelGGL((deform_tile3_spill_kernel<T, ORDER>), dim3(nsp), dim3(kBlock), 0, stream,
                           g, ve, tg);
        e = hipGetLastError();
    }
    const hipError_t e2 = hipFreeAsync(spill, stream);
    return e != hipSuccess ? e : e2;
}

}  // namespace

bool deform_tile_supported(const GridGeom& g, const IOView& v, int gradient)
{
    if (gradient || g.naxis != 3 || v.order < 2)
        return false;
    if (!deform_fast_supported(g, v, gradient))
        return false;
    for (int k = 0; k < 3; ++k)
        if (g.in_len[k] >= 0x3fffffff || g.out_len[k] >= 0x3fffffff || g.ncp[k] > 4096)
            return false;
    // D + P + Q live in the LDS region that later holds the source box; keep the block <= 48 KiB
    if (dpq_bytes(g) > (size_t)47 * 1024)
        return false;
    return true;
}

hipError_t launch_deform_tile(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream)
{
    if (!deform_tile_supported(g, v, gradient))
        return hipErrorNotSupported;
    if (v.in_dtype == EDHIP_F32) {
        switch (v.order) {
        case 2: return launch_tile<float, 2, false>(g, v, stream);
        case 3: return launch_tile<float, 3, true>(g, v, stream);
        case 4: return launch_tile<float, 4, false>(g, v, stream);
        case 5: return launch_tile<float, 5, true>(g, v, stream);
        default: return hipErrorNotSupported;
        }
    }
    switch (v.order) {
    case 2: return launch_tile<double, 2, false>(g, v, stream);
    case 3: return launch_tile<double, 3, false>(g, v, stream);
    case 4: return launch_tile<double, 4, false>(g, v, stream);
    case 5: return launch_tile<double, 5, false>(g, v, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace ed
