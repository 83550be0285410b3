// tools/experiments/deform_k2z.hip -- PROFILING BUILD ONLY (make EXPERIMENTS=1; measured slower than hot_grad_kernel,
// profiles/r06_k2_zwalk.txt).  A K2 of round 6 that did not ship: the gradient scatter-add of the benchmark case (float32 volumes, 3 deformed axes, unit
// stride along x on both sides, spline orders 1-3; deform.c:926-997) on the tables of the z-walk forward route
// (deform_k1z.hip).  The scatter itself is hot_grad_kernel's (deform_hot.hip): tiles of 8 (z) x 8 (y) x 16 (x) output
// voxels, four voxels per lane, taps scattered into fixed-point LDS cells with integer atomics (per-tile scale from the
// tile's sum of |dY|: a rigorous no-overflow bound), one float atomic per touched source element in the flush.  What is
// different:
//
//   * A workgroup walks a strip of tiles along z.  A lane keeps its (y, x) column for the whole strip, so the four
//     control planes x three components of R[o_y][o_x][k_z][c] it needs are lane-constant registers, and a wave works on
//     one z slice at a time, so the z weights are scalar loads: a voxel's displacement costs no LDS access.  (The strips
//     of hot_grad_kernel run along x: every voxel read 4 control columns x 3 components of its Q row and its x-table
//     entry from LDS -- 11 of the kernel's 81 LDS instructions per 64 voxels, on the pipe that bounds it -- and every
//     strip began by copying 10 KB of Q rows and 3 KB of x table into LDS.)
//   * The LDS a workgroup owns is cells and 200 bytes: 24 KiB + 0.2 instead of 37.6 KiB.
//   * A small argument block; what the rare paths read (boundary map, affine map) sits in the geometry buffer (ZGen) and
//     is loaded where it is used (hot_grad_kernel: 79 scalar registers spilled to vector lanes).
//
// The tables (R, the z table, ZGen) come from the geometry kernel of deform_k1z.hip in its tables-only form -- the launch
// that also clears the gradient accumulators (EDHIP_FLAG_ZERO_GRADIENT) on spare workgroups.  Tiles whose cells do not
// fit are taken as their two x halves (with the forward call's boxes) or scatter straight to global memory: this kernel
// always serves itself, so it is launched only for geometries whose recent calls left (almost) nothing to the spill
// lists -- everything else stays on hot_grad_kernel and the spill levels behind it.
//
// Bits: coordinates come from R like the forward route's, so a voxel's window start and fractions are the forward
// kernel's, bit for bit; the fixed-point accumulation and its bounds are hot_grad_kernel's (include/edhip.h).
#include <hip/hip_runtime.h>

#include <cstring>

#include "ed_device.h"
#include "ed_hot.h"
#include "ed_params.h"
#include "ed_tile.h"
#include "ed_zwalk.h"

namespace ed {
namespace tile {

namespace {

struct ZGrad {
    float* dx;
    const float* dy;
    const double* r;
    const AxTab* zt;
    const int* boxes;         // the forward call's tile boxes (EDHIP_FLAG_USE_BOXES) or nullptr
    const long long* steps;   // STEPS: [nsteps][2] element offsets (volume, image)
    int* hint;
    long long vol_bstride, img_bstride, r_bstride;
    int vol_sy, vol_sz, img_sy, img_sz;
    int box_cap, small_cap;
    int tiles_z, tiles_y, tiles_x, tiles_x2, ntiles;      // tiles_x: 8-wide (the boxes' index), tiles_x2: 16-wide
    int strip_tiles, nstrips, total_strips;
    int rcol_bytes, out_z, out_y, out_x;
    int in_len[3], off[3];
    int nsteps, deal, dbg;
};

constexpr int kGOffRed = 0;           // int[3][8]: lo[3], hi[3], -, - (triple-buffered)
constexpr int kGOffSum = 96;          // float[2][4]: per-wave sum |dY|
constexpr int kGOffBox = 128;         // cells

template <int ORDER, bool AFFINE, bool STEPS>
__global__ __launch_bounds__(kBlock, 4) void k2z_grad_kernel(const ZGrad a, czgen_p zn)
{
    constexpr int NT = ORDER + 1;
    constexpr int TX = 16, NV = 4, ZSTEP = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    // strips (runs of 8 x 8 x 16 tiles along z at one (ty, tx2)) dealt to the XCDs in skewed chunks, like the forward's
    int sample, ty, tx2, tz0, ntile;
    {
        const int b = blockIdx.x;
        const int x = b & 7, j = b >> 3;
        const int r = j / a.deal;
        int s = (r * 8 + ((x + r) & 7)) * a.deal + (j - r * a.deal);
        if (s >= a.total_strips)
            return;
        sample = s / a.nstrips;
        s -= sample * a.nstrips;
        tx2 = s % a.tiles_x2;
        s /= a.tiles_x2;
        ty = s % a.tiles_y;
        tz0 = (s / a.tiles_y) * a.strip_tiles;
        ntile = min(a.strip_tiles, a.tiles_z - tz0);
    }
    // the accumulator cells start at zero and every flush leaves the cells it read at zero again
    for (int e = tid * 4; e < a.box_cap; e += kBlock * 4)
        *reinterpret_cast<int4*>(smem + kGOffBox + e * 4) = make_int4(0, 0, 0, 0);
    int* sred = reinterpret_cast<int*>(smem + kGOffRed);
    if (tid < 24)
        sred[tid] = (tid & 7) < 3 ? 0x7fffffff : (int)0x80000000;
    __syncthreads();
    int* box = reinterpret_cast<int*>(smem + kGOffBox);

    const int lane = tid & 63;
    const int wave = uni(tid >> 6);
    // lane -> voxel: the 16 lanes that go through the LDS together hold voxels two apart along x and y (see deform_hot.hip)
    const int zq = wave >> 1;
    const int xx = 2 * (tid & 7) + ((tid >> 4) & 1);
    const int yy = 4 * ((tid >> 6) & 1) + ((tid >> 5) & 1) + 2 * ((tid >> 3) & 1);
    float* dx = a.dx + sample * a.vol_bstride;
    const float* __restrict__ dy = a.dy + sample * a.img_bstride;
    const int oy = ty * kT + yy, ox = tx2 * TX + xx;
    const bool vyx = oy < a.out_y && ox < a.out_x;
    const int vol_sz = a.vol_sz, vol_sy = a.vol_sy;
    const int ostep = ZSTEP * a.img_sz;
    const char* rcol = reinterpret_cast<const char*>(a.r + sample * a.r_bstride) +
                       ((size_t)min(oy, a.out_y - 1) * a.out_x + min(ox, a.out_x - 1)) * (size_t)a.rcol_bytes;
    cdbl_p zt = (cdbl_p)(const void*)a.zt;
    cll_p steps = (cll_p)(const void*)a.steps;
    const int nsteps = STEPS ? a.nsteps : 1;
    double Pyx[3] = {0.0, 0.0, 0.0};      // affine: A[h][1] oy + A[h][2] ox + A[h][3] + off_h
    if (AFFINE) {
#pragma unroll
        for (int h = 0; h < 3; ++h)
            Pyx[h] = fma(zn->aff[h * 4 + 2], (double)ox, fma(zn->aff[h * 4 + 1], (double)oy, zn->aff[h * 4 + 3] + zn->offd[h]));
    }
    ZTaps tp;
    tp.key[0] = tp.key[1] = tp.key[2] = tp.key[3] = -1;

    // dY of a tile's first step is loaded one tile ahead
    float gnext[NV];
    {
        const int oz0 = tz0 * kT + zq;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            gnext[i] = (vyx && oz0 + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + (long long)(oz0 + ZSTEP * i) * a.img_sz + oy * a.img_sy + ox] : 0.f;
    }
    int phase = 0;
    // half: -1 = the whole 16-wide tile; 0 / 1 = an oversize tile taken again as its x halves
    int half = -1, half_next = -1;
    for (int ti = 0; ti < ntile; half = half_next, half_next = half == 0 ? 1 : -1, ti += half < 0 ? 1 : 0) {
        int* red = sred + (ti % 3) * 8;
        const int tz = tz0 + ti;
        const int oz0 = tz * kT + zq;
        const long long ooff0 = (long long)oz0 * a.img_sz + oy * a.img_sy + ox;
        float gpre[NV];
        if (half >= 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gpre[i] = (vyx && oz0 + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + ooff0 + i * ostep] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gpre[i] = gnext[i];
        }
        if (half < 0 && ti + 1 < ntile) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gnext[i] = (vyx && oz0 + kT + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + ooff0 + (long long)kT * a.img_sz + i * ostep] : 0.f;
        }
        // the forward call's boxes: requested here, ahead of the barrier
        const bool given = a.boxes != nullptr;
        int gb0[3] = {0, 0, 0}, gbhi[3] = {-1, -1, -1};
        if (given) {
            const size_t t0 = (size_t)sample * a.ntiles + ((size_t)tz * a.tiles_y + ty) * a.tiles_x + 2 * tx2;
            cint_p bx = (cint_p)(const void*)a.boxes + t0 * 8;
            const bool two = 2 * tx2 + 1 < a.tiles_x;
            if (half < 0) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    gb0[h] = two ? min(bx[h], bx[8 + h]) : bx[h];
                    gbhi[h] = two ? max(bx[3 + h], bx[8 + 3 + h]) : bx[3 + h];
                }
            } else {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    gb0[h] = bx[half * 8 + h];
                    gbhi[h] = bx[half * 8 + 3 + h];
                }
            }
        }
        // One voxel at a time, nothing kept per voxel (see deform_hot.hip): window start and fractions of voxel i
        auto voxel = [&](int i, int* start, float* frac) -> bool {
            const int oz = oz0 + ZSTEP * i;
            double zw[4];
            k1z_slice(zt, rcol, min(oz, a.out_z - 1), tp, zw);
            double d[3];
            k1z_disp(tp, zw, d);
            const int b[3] = {oz + a.off[0], oy + a.off[1], ox + a.off[2]};
            double P[3] = {0.0, 0.0, 0.0};
            if (AFFINE) {
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    P[h] = fma(zn->aff[h * 4 + 0], (double)oz, Pyx[h]);
            }
            const bool cst = k1z_coords<ORDER, AFFINE>(zn, d, b, P, start, frac);
            // constant voxels contribute nothing (:928); in a half pass the other half's lanes sit out
            return vyx && oz < a.out_z && !cst && (half < 0 || (xx >> 3) == half);
        };
        int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
        if (!given) {
#pragma unroll 1
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
                gval[i] = gpre[i];
                // inf / NaN gradients have no fixed-point scale: left out of the sum, scattered with float atomics below
                gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
            }
            gm = wave_sum(gm);
            if (lane == 0)
                reinterpret_cast<float*>(smem + kGOffSum)[(phase & 1) * 4 + wave] = gm;
        }
        if (!given)
            box_reduce_to_lds(red, lane, lo, hi);
        lds_barrier();     // B1: box and sum known; the previous tile's flush is done (cells back at zero)
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
            // an empty box may be a stale one: keep going with zero cells -- every live voxel then fails the window test
            // below and is scattered directly
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
        if (given && ((unsigned)ext[0] > 4096u || (unsigned)ext[1] > 4096u))
            pitch = 0;          // (a handed-over box is not trusted with the products below)
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const int nbox = nrows * pitch;
        if (a.hint && tid == 0 && half < 0 && (pitch == 0 || nbox > a.small_cap))
            atomicAdd(a.hint, TX / kT);   // spill feedback, in 8-wide tiles
        // a tile that does not fit keeps an EMPTY box -- every live voxel then fails the window test below and scatters
        // its taps straight to global memory (the path of a stale handed-over box)
        bool direct_tile = false;
        if (pitch == 0 || nbox > a.box_cap) {
            if (given && half < 0 && 2 * tx2 + 1 < a.tiles_x) {
                half_next = 0;       // taken again as its two x halves, each with the box of the forward tile it is
                continue;
            }
            direct_tile = true;
        }
        const bool interior = b0[0] >= 0 && b0[0] + ext[0] <= a.in_len[0] && b0[1] >= 0 && b0[1] + ext[1] <= a.in_len[1] &&
                              b0[2] >= 0 && b0[2] + ext[2] <= a.in_len[2];

        for (int ss = 0; ss < nsteps; ++ss, ++phase) {
            float* dst = dx + (STEPS ? steps[2 * ss] : 0);
            float* gsum = reinterpret_cast<float*>(smem + kGOffSum) + (phase & 1) * 4;
            if (ss > 0) {
                // later steps (channels) of the same tile: their own sum, after the previous flush
                float gm = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const bool inb = vyx && oz0 + ZSTEP * i < a.out_z;
                    gval[i] = inb ? dy[steps[2 * ss + 1] + ooff0 + i * ostep] : 0.f;
                    gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
                }
                gm = wave_sum(gm);
                if (lane == 0)
                    gsum[wave] = gm;
                lds_barrier();           // sum known; the previous step's flush is done with the box
            }
            const float gtot = unif((gsum[0] + gsum[1]) + (gsum[2] + gsum[3]));
            // |sum in a cell| <= max tap weight * sum over the tile of |dY|: this scale cannot overflow
            constexpr float kC = (float)((2147483648.0 - 1024.0) / ((ORDER == 1 ? 1.0 : ORDER == 2 ? 0.4219 : 0.2963) * 1.001));
            const float scale = gtot > 0.f ? fminf(kC * __frcp_rn(gtot), 3.0e38f) : 0.f;
            const float inv_scale = gtot > 0.f ? __frcp_rn(scale) : 0.f;

#pragma unroll 1
            for (int i = 0; i < NV; ++i) {
                float gv = gval[0];
#pragma unroll
                for (int k = 1; k < NV; ++k)
                    gv = i == k ? gval[k] : gv;
                if (gv == 0.f)
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
                if ((__float_as_int(gv) & 0x7f800000) == 0x7f800000 || outside) {
                    // inf / NaN gradient (no fixed-point scale), or a window outside a stale box: float atomics straight
                    // to global memory (rare, rolled loop)
#pragma unroll 1
                    for (int t = 0; t < NT * NT * NT; ++t) {
                        const int l0 = t / (NT * NT), l1 = (t / NT) % NT, l2 = t % NT;
                        const int zs = mirror_i32(st[0] + l0, a.in_len[0]);
                        const int ys = mirror_i32(st[1] + l1, a.in_len[1]);
                        const int xs = mirror_i32(st[2] + l2, a.in_len[2]);
                        float wp = w0[0], wq = w1[0], wr = w2[0];
#pragma unroll
                        for (int l = 1; l < NT; ++l) {
                            wp = l0 == l ? w0[l] : wp;
                            wq = l1 == l ? w1[l] : wq;
                            wr = l2 == l ? w2[l] : wr;
                        }
                        unsafeAtomicAdd(dst + (zs * vol_sz + ys * vol_sy + xs), gv * wp * wq * wr);
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
                        for (int l2 = 0; l2 < NT; ++l2) {
                            atomicAdd(reinterpret_cast<unsigned*>(rp + l2), (unsigned)round_half_up_i32(pr[l2]));
                        }
                    }
                }
            }
            lds_barrier();               // B3: all contributions are in
            // flush: half a wave per box row, lanes along x -- one float atomic per touched source element, runs of
            // consecutive addresses (deform.c:791-813: mirror-mapped at the edges); see deform_hot.hip
            {
                constexpr int FL = 32;                         // lanes per box row
                constexpr int FR = kBlock / FL;                // rows per pass
                constexpr int FU = 4;                          // rows in flight per lane
                const int sub = tid & (FL - 1);
                const int rslot = tid / FL;
                const float inv_by = 1.f / (float)by;
                const int nr = direct_tile ? 0 : nrows;
                const int dz8 = (int)(((float)FR + 0.5f) * inv_by), dy8 = FR - dz8 * by;
                const int step8 = dz8 * vol_sz + dy8 * vol_sy, wrapfix = vol_sz - by * vol_sy;
                for (int xo = 0; xo < ext[2]; xo += FL) {
                    const int xi = xo + sub;
                    const bool xin = xi < ext[2];
                    const int xs = interior ? xi : mirror_i32(b0[2] + xi, a.in_len[2]);
                    int zr = (int)(((float)rslot + 0.5f) * inv_by), yr = rslot - zr * by;
                    int rowoff = (b0[0] + zr) * vol_sz + (b0[1] + yr) * vol_sy + b0[2];
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
                            if (acc[k] != 0) {
                                int off = rowoff;
                                if (!interior)
                                    off = mirror_i32(b0[0] + zr, a.in_len[0]) * vol_sz + mirror_i32(b0[1] + yr, a.in_len[1]) * vol_sy;
                                unsafeAtomicAdd(dst + (off + xs), (float)acc[k] * inv_scale);
                            }
                            // the next row of this lane
                            yr += dy8;
                            zr += dz8;
                            rowoff += step8;
                            if (yr >= by) {
                                yr -= by;
                                zr += 1;
                                rowoff += wrapfix;
                            }
                        }
                    }
                }
            }
        }
    }
}


// ---- the same kernel on 8 x 8 x 8 tiles, two waves per workgroup: up to eight independent workgroups per CU ----
template <int ORDER, bool AFFINE, bool STEPS>
__global__ __launch_bounds__(128, 2) void k2y_grad_kernel(const ZGrad a, czgen_p zn)
{
    constexpr int NT = ORDER + 1;
    constexpr int TX = 8, NV = 4, ZSTEP = 2;
    constexpr int kBlock = 128;        // (shadows the 256 of the tile kernels: two waves per workgroup)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    // strips (runs of 8 x 8 x 16 tiles along z at one (ty, tx2)) dealt to the XCDs in skewed chunks, like the forward's
    int sample, ty, tx2, tz0, ntile;
    {
        const int b = blockIdx.x;
        const int x = b & 7, j = b >> 3;
        const int r = j / a.deal;
        int s = (r * 8 + ((x + r) & 7)) * a.deal + (j - r * a.deal);
        if (s >= a.total_strips)
            return;
        sample = s / a.nstrips;
        s -= sample * a.nstrips;
        tx2 = s % a.tiles_x;
        s /= a.tiles_x;
        ty = s % a.tiles_y;
        tz0 = (s / a.tiles_y) * a.strip_tiles;
        ntile = min(a.strip_tiles, a.tiles_z - tz0);
    }
    // the accumulator cells start at zero and every flush leaves the cells it read at zero again
    for (int e = tid * 4; e < a.box_cap; e += kBlock * 4)
        *reinterpret_cast<int4*>(smem + kGOffBox + e * 4) = make_int4(0, 0, 0, 0);
    int* sred = reinterpret_cast<int*>(smem + kGOffRed);
    if (tid < 24)
        sred[tid] = (tid & 7) < 3 ? 0x7fffffff : (int)0x80000000;
    __syncthreads();
    int* box = reinterpret_cast<int*>(smem + kGOffBox);

    const int lane = tid & 63;
    const int wave = uni(tid >> 6);
    // lane -> voxel: the 16 lanes that go through the LDS together hold voxels two apart along x and y (see deform_hot.hip)
    // (one wave per z slice; its 16-lane groups hold voxels two apart along x and y)
    const int zq = wave;
    const int xx = 2 * (lane & 3) + ((lane >> 4) & 1);
    const int yy = 2 * ((lane >> 2) & 3) + ((lane >> 5) & 1);
    float* dx = a.dx + sample * a.vol_bstride;
    const float* __restrict__ dy = a.dy + sample * a.img_bstride;
    const int oy = ty * kT + yy, ox = tx2 * TX + xx;
    const bool vyx = oy < a.out_y && ox < a.out_x;
    const int vol_sz = a.vol_sz, vol_sy = a.vol_sy;
    const int ostep = ZSTEP * a.img_sz;
    const char* rcol = reinterpret_cast<const char*>(a.r + sample * a.r_bstride) +
                       ((size_t)min(oy, a.out_y - 1) * a.out_x + min(ox, a.out_x - 1)) * (size_t)a.rcol_bytes;
    cdbl_p zt = (cdbl_p)(const void*)a.zt;
    cll_p steps = (cll_p)(const void*)a.steps;
    const int nsteps = STEPS ? a.nsteps : 1;
    double Pyx[3] = {0.0, 0.0, 0.0};      // affine: A[h][1] oy + A[h][2] ox + A[h][3] + off_h
    if (AFFINE) {
#pragma unroll
        for (int h = 0; h < 3; ++h)
            Pyx[h] = fma(zn->aff[h * 4 + 2], (double)ox, fma(zn->aff[h * 4 + 1], (double)oy, zn->aff[h * 4 + 3] + zn->offd[h]));
    }
    ZTaps tp;
    tp.key[0] = tp.key[1] = tp.key[2] = tp.key[3] = -1;

    // dY of a tile's first step is loaded one tile ahead
    float gnext[NV];
    {
        const int oz0 = tz0 * kT + zq;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            gnext[i] = (vyx && oz0 + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + (long long)(oz0 + ZSTEP * i) * a.img_sz + oy * a.img_sy + ox] : 0.f;
    }
    int phase = 0;
    // half: -1 = the whole 16-wide tile; 0 / 1 = an oversize tile taken again as its x halves
    int half = -1, half_next = -1;
    for (int ti = 0; ti < ntile; half = half_next, half_next = half == 0 ? 1 : -1, ti += half < 0 ? 1 : 0) {
        int* red = sred + (ti % 3) * 8;
        const int tz = tz0 + ti;
        const int oz0 = tz * kT + zq;
        const long long ooff0 = (long long)oz0 * a.img_sz + oy * a.img_sy + ox;
        float gpre[NV];
        if (half >= 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gpre[i] = (vyx && oz0 + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + ooff0 + i * ostep] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gpre[i] = gnext[i];
        }
        if (half < 0 && ti + 1 < ntile) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gnext[i] = (vyx && oz0 + kT + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + ooff0 + (long long)kT * a.img_sz + i * ostep] : 0.f;
        }
        // the forward call's boxes: requested here, ahead of the barrier
        const bool given = a.boxes != nullptr;
        int gb0[3] = {0, 0, 0}, gbhi[3] = {-1, -1, -1};
        if (given) {
            const size_t t0 = (size_t)sample * a.ntiles + ((size_t)tz * a.tiles_y + ty) * a.tiles_x + tx2;
            cint_p bx = (cint_p)(const void*)a.boxes + t0 * 8;
            const bool two = false;
            if (half < 0) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    gb0[h] = two ? min(bx[h], bx[8 + h]) : bx[h];
                    gbhi[h] = two ? max(bx[3 + h], bx[8 + 3 + h]) : bx[3 + h];
                }
            } else {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    gb0[h] = bx[half * 8 + h];
                    gbhi[h] = bx[half * 8 + 3 + h];
                }
            }
        }
        // One voxel at a time, nothing kept per voxel (see deform_hot.hip): window start and fractions of voxel i
        auto voxel = [&](int i, int* start, float* frac) -> bool {
            const int oz = oz0 + ZSTEP * i;
            double zw[4];
            k1z_slice(zt, rcol, min(oz, a.out_z - 1), tp, zw);
            double d[3];
            k1z_disp(tp, zw, d);
            const int b[3] = {oz + a.off[0], oy + a.off[1], ox + a.off[2]};
            double P[3] = {0.0, 0.0, 0.0};
            if (AFFINE) {
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    P[h] = fma(zn->aff[h * 4 + 0], (double)oz, Pyx[h]);
            }
            const bool cst = k1z_coords<ORDER, AFFINE>(zn, d, b, P, start, frac);
            // constant voxels contribute nothing (:928); in a half pass the other half's lanes sit out
            return vyx && oz < a.out_z && !cst;
        };
        int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
        if (!given) {
#pragma unroll 1
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
                gval[i] = gpre[i];
                // inf / NaN gradients have no fixed-point scale: left out of the sum, scattered with float atomics below
                gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
            }
            gm = wave_sum(gm);
            if (lane == 0)
                reinterpret_cast<float*>(smem + kGOffSum)[(phase & 1) * 4 + wave] = gm;
        }
        if (!given)
            box_reduce_to_lds(red, lane, lo, hi);
        lds_barrier();     // B1: box and sum known; the previous tile's flush is done (cells back at zero)
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
            // an empty box may be a stale one: keep going with zero cells -- every live voxel then fails the window test
            // below and is scattered directly
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
        if (given && ((unsigned)ext[0] > 4096u || (unsigned)ext[1] > 4096u))
            pitch = 0;          // (a handed-over box is not trusted with the products below)
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const int nbox = nrows * pitch;
        if (a.hint && tid == 0 && half < 0 && (pitch == 0 || nbox > a.small_cap))
            atomicAdd(a.hint, 1);   // spill feedback, in 8-wide tiles
        // a tile that does not fit keeps an EMPTY box -- every live voxel then fails the window test below and scatters
        // its taps straight to global memory (the path of a stale handed-over box)
        bool direct_tile = false;
        if (pitch == 0 || nbox > a.box_cap) {
            if (false) {
                half_next = 0;       // taken again as its two x halves, each with the box of the forward tile it is
                continue;
            }
            direct_tile = true;
        }
        const bool interior = b0[0] >= 0 && b0[0] + ext[0] <= a.in_len[0] && b0[1] >= 0 && b0[1] + ext[1] <= a.in_len[1] &&
                              b0[2] >= 0 && b0[2] + ext[2] <= a.in_len[2];

        for (int ss = 0; ss < nsteps; ++ss, ++phase) {
            float* dst = dx + (STEPS ? steps[2 * ss] : 0);
            float* gsum = reinterpret_cast<float*>(smem + kGOffSum) + (phase & 1) * 4;
            if (ss > 0) {
                // later steps (channels) of the same tile: their own sum, after the previous flush
                float gm = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const bool inb = vyx && oz0 + ZSTEP * i < a.out_z;
                    gval[i] = inb ? dy[steps[2 * ss + 1] + ooff0 + i * ostep] : 0.f;
                    gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
                }
                gm = wave_sum(gm);
                if (lane == 0)
                    gsum[wave] = gm;
                lds_barrier();           // sum known; the previous step's flush is done with the box
            }
            const float gtot = unif(gsum[0] + gsum[1]);
            // |sum in a cell| <= max tap weight * sum over the tile of |dY|: this scale cannot overflow
            constexpr float kC = (float)((2147483648.0 - 1024.0) / ((ORDER == 1 ? 1.0 : ORDER == 2 ? 0.4219 : 0.2963) * 1.001));
            const float scale = gtot > 0.f ? fminf(kC * __frcp_rn(gtot), 3.0e38f) : 0.f;
            const float inv_scale = gtot > 0.f ? __frcp_rn(scale) : 0.f;

#pragma unroll 1
            for (int i = 0; i < NV; ++i) {
                float gv = gval[0];
#pragma unroll
                for (int k = 1; k < NV; ++k)
                    gv = i == k ? gval[k] : gv;
                if (gv == 0.f)
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
                if ((__float_as_int(gv) & 0x7f800000) == 0x7f800000 || outside) {
                    // inf / NaN gradient (no fixed-point scale), or a window outside a stale box: float atomics straight
                    // to global memory (rare, rolled loop)
#pragma unroll 1
                    for (int t = 0; t < NT * NT * NT; ++t) {
                        const int l0 = t / (NT * NT), l1 = (t / NT) % NT, l2 = t % NT;
                        const int zs = mirror_i32(st[0] + l0, a.in_len[0]);
                        const int ys = mirror_i32(st[1] + l1, a.in_len[1]);
                        const int xs = mirror_i32(st[2] + l2, a.in_len[2]);
                        float wp = w0[0], wq = w1[0], wr = w2[0];
#pragma unroll
                        for (int l = 1; l < NT; ++l) {
                            wp = l0 == l ? w0[l] : wp;
                            wq = l1 == l ? w1[l] : wq;
                            wr = l2 == l ? w2[l] : wr;
                        }
                        unsafeAtomicAdd(dst + (zs * vol_sz + ys * vol_sy + xs), gv * wp * wq * wr);
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
                        for (int l2 = 0; l2 < NT; ++l2) {
                            atomicAdd(reinterpret_cast<unsigned*>(rp + l2), (unsigned)round_half_up_i32(pr[l2]));
                        }
                    }
                }
            }
            lds_barrier();               // B3: all contributions are in
            // flush: half a wave per box row, lanes along x -- one float atomic per touched source element, runs of
            // consecutive addresses (deform.c:791-813: mirror-mapped at the edges); see deform_hot.hip
            {
                constexpr int FL = 16;                         // lanes per box row
                constexpr int FR = kBlock / FL;                // rows per pass
                constexpr int FU = 4;                          // rows in flight per lane
                const int sub = tid & (FL - 1);
                const int rslot = tid / FL;
                const float inv_by = 1.f / (float)by;
                const int nr = direct_tile ? 0 : nrows;
                const int dz8 = (int)(((float)FR + 0.5f) * inv_by), dy8 = FR - dz8 * by;
                const int step8 = dz8 * vol_sz + dy8 * vol_sy, wrapfix = vol_sz - by * vol_sy;
                for (int xo = 0; xo < ext[2]; xo += FL) {
                    const int xi = xo + sub;
                    const bool xin = xi < ext[2];
                    const int xs = interior ? xi : mirror_i32(b0[2] + xi, a.in_len[2]);
                    int zr = (int)(((float)rslot + 0.5f) * inv_by), yr = rslot - zr * by;
                    int rowoff = (b0[0] + zr) * vol_sz + (b0[1] + yr) * vol_sy + b0[2];
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
                            if (acc[k] != 0) {
                                int off = rowoff;
                                if (!interior)
                                    off = mirror_i32(b0[0] + zr, a.in_len[0]) * vol_sz + mirror_i32(b0[1] + yr, a.in_len[1]) * vol_sy;
                                unsafeAtomicAdd(dst + (off + xs), (float)acc[k] * inv_scale);
                            }
                            // the next row of this lane
                            yr += dy8;
                            zr += dz8;
                            rowoff += step8;
                            while (yr >= by) {
                                yr -= by;
                                zr += 1;
                                rowoff += wrapfix;
                            }
                        }
                    }
                }
            }
        }
    }
}


// ================================================================================================
// K2s: producer / consumer waves
// ================================================================================================
// The same tile, the same arithmetic, the same cells -- but the two halves of a voxel's work run in DIFFERENT waves.
// Measured on the kernel above (256^3, sigma 5, whole gradient call 326 us): without the 64 products / conversions /
// LDS atomics of a voxel 173 us, without any per-voxel work 113 us.  The scatter is bound by the LDS instruction path
// (a ds_add_u32 moves two dwords per lane: 4 cycles per wave-instruction and CU, 64 per voxel: ~110 us of the launch at
// the clock the chip holds), the coordinates by the vector ALU (fp64 sums, floor, conversions), and a wave does one after
// the other: while it scatters its VALU slots idle, while it computes coordinates the LDS path idles, and the barriers
// of a tile keep the four waves of a workgroup in the same phase.  Here a workgroup has eight waves: waves 0-3 (producers)
// compute window starts, fractions and the scaled gradient of their voxels and leave them in a small LDS ring; waves
// 4-7 (consumers) turn a record into weights and 64 fixed-point atomics.  One barrier per round of 256 voxels; the
// producers of round r run beside the consumers of round r - 1.
constexpr int kSOffRed = 0;           // int[3][8]
constexpr int kSOffSum = 96;          // float[2][4]
constexpr int kSOffRing = 128;        // [2 slots][4 waves][5 fields][64 lanes] dwords
constexpr int kSRingBytes = 2 * 4 * 5 * 64 * 4;
constexpr int kSOffBox = kSOffRing + kSRingBytes;
constexpr int kSBlock = 512;

template <int ORDER, bool AFFINE, bool STEPS>
__global__ __launch_bounds__(kSBlock, 2) void k2s_grad_kernel(const ZGrad a, czgen_p zn)
{
    constexpr int NT = ORDER + 1;
    constexpr int TX = 16, NV = 4, ZSTEP = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int sample, ty, tx2, tz0, ntile;
    {
        const int b = blockIdx.x;
        const int x = b & 7, j = b >> 3;
        const int r = j / a.deal;
        int s = (r * 8 + ((x + r) & 7)) * a.deal + (j - r * a.deal);
        if (s >= a.total_strips)
            return;
        sample = s / a.nstrips;
        s -= sample * a.nstrips;
        tx2 = s % a.tiles_x2;
        s /= a.tiles_x2;
        ty = s % a.tiles_y;
        tz0 = (s / a.tiles_y) * a.strip_tiles;
        ntile = min(a.strip_tiles, a.tiles_z - tz0);
    }
    for (int e = tid * 4; e < a.box_cap; e += kSBlock * 4)
        *reinterpret_cast<int4*>(smem + kSOffBox + e * 4) = make_int4(0, 0, 0, 0);
    int* sred = reinterpret_cast<int*>(smem + kSOffRed);
    if (tid < 24)
        sred[tid] = (tid & 7) < 3 ? 0x7fffffff : (int)0x80000000;
    __syncthreads();
    int* box = reinterpret_cast<int*>(smem + kSOffBox);
    int* ring = reinterpret_cast<int*>(smem + kSOffRing);

    const int lane = tid & 63;
    const int wave = uni(tid >> 6);
    const bool producer = wave < 4;
    const int pw = wave & 3;              // the producer wave this wave is / consumes
    const int pt = tid & 255;             // (producers) the thread's place among the 256 voxels of a round
    const int zq = pw >> 1;
    const int xx = 2 * (pt & 7) + ((pt >> 4) & 1);
    const int yy = 4 * ((pt >> 6) & 1) + ((pt >> 5) & 1) + 2 * ((pt >> 3) & 1);
    float* dx = a.dx + sample * a.vol_bstride;
    const float* __restrict__ dy = a.dy + sample * a.img_bstride;
    const int oy = ty * kT + yy, ox = tx2 * TX + xx;
    const bool vyx = oy < a.out_y && ox < a.out_x;
    const int vol_sz = a.vol_sz, vol_sy = a.vol_sy;
    const int ostep = ZSTEP * a.img_sz;
    const char* rcol = reinterpret_cast<const char*>(a.r + sample * a.r_bstride) +
                       ((size_t)min(oy, a.out_y - 1) * a.out_x + min(ox, a.out_x - 1)) * (size_t)a.rcol_bytes;
    cdbl_p zt = (cdbl_p)(const void*)a.zt;
    cll_p steps = (cll_p)(const void*)a.steps;
    const int nsteps = STEPS ? a.nsteps : 1;
    double Pyx[3] = {0.0, 0.0, 0.0};
    if (AFFINE && producer) {
#pragma unroll
        for (int h = 0; h < 3; ++h)
            Pyx[h] = fma(zn->aff[h * 4 + 2], (double)ox, fma(zn->aff[h * 4 + 1], (double)oy, zn->aff[h * 4 + 3] + zn->offd[h]));
    }
    ZTaps tp;
    tp.key[0] = tp.key[1] = tp.key[2] = tp.key[3] = -1;

    float gnext[NV] = {0.f, 0.f, 0.f, 0.f};
    if (producer) {
        const int oz0 = tz0 * kT + zq;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            gnext[i] = (vyx && oz0 + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + (long long)(oz0 + ZSTEP * i) * a.img_sz + oy * a.img_sy + ox] : 0.f;
    }
    int phase = 0;
    int half = -1, half_next = -1;
    for (int ti = 0; ti < ntile; half = half_next, half_next = half == 0 ? 1 : -1, ti += half < 0 ? 1 : 0) {
        int* red = sred + (ti % 3) * 8;
        const int tz = tz0 + ti;
        const int oz0 = tz * kT + zq;
        const long long ooff0 = (long long)oz0 * a.img_sz + oy * a.img_sy + ox;
        float gval[NV] = {0.f, 0.f, 0.f, 0.f};
        if (producer) {
            if (half >= 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    gval[i] = (vyx && oz0 + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + ooff0 + i * ostep] : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    gval[i] = gnext[i];
            }
            if (half < 0 && ti + 1 < ntile) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    gnext[i] = (vyx && oz0 + kT + ZSTEP * i < a.out_z) ? dy[(STEPS ? steps[1] : 0) + ooff0 + (long long)kT * a.img_sz + i * ostep] : 0.f;
            }
        }
        const bool given = a.boxes != nullptr;
        int gb0[3] = {0, 0, 0}, gbhi[3] = {-1, -1, -1};
        if (given) {
            const size_t t0 = (size_t)sample * a.ntiles + ((size_t)tz * a.tiles_y + ty) * a.tiles_x + 2 * tx2;
            cint_p bx = (cint_p)(const void*)a.boxes + t0 * 8;
            const bool two = 2 * tx2 + 1 < a.tiles_x;
            if (half < 0) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    gb0[h] = two ? min(bx[h], bx[8 + h]) : bx[h];
                    gbhi[h] = two ? max(bx[3 + h], bx[8 + 3 + h]) : bx[3 + h];
                }
            } else {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    gb0[h] = bx[half * 8 + h];
                    gbhi[h] = bx[half * 8 + 3 + h];
                }
            }
        }
        // (producers) window start and fractions of voxel i
        auto voxel = [&](int i, int* start, float* frac) -> bool {
            const int oz = oz0 + ZSTEP * i;
            double zw[4];
            k1z_slice(zt, rcol, min(oz, a.out_z - 1), tp, zw);
            double d[3];
            k1z_disp(tp, zw, d);
            const int b[3] = {oz + a.off[0], oy + a.off[1], ox + a.off[2]};
            double P[3] = {0.0, 0.0, 0.0};
            if (AFFINE) {
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    P[h] = fma(zn->aff[h * 4 + 0], (double)oz, Pyx[h]);
            }
            const bool cst = k1z_coords<ORDER, AFFINE>(zn, d, b, P, start, frac);
            return vyx && oz < a.out_z && !cst && (half < 0 || (xx >> 3) == half);
        };
        if (producer) {
            if (!given) {
                int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
                int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
#pragma unroll 1
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
                box_reduce_to_lds(red, lane, lo, hi);
            }
            float gm = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
            gm = wave_sum(gm);
            if (lane == 0)
                reinterpret_cast<float*>(smem + kSOffSum)[(phase & 1) * 4 + pw] = gm;
        }
        lds_barrier();     // A: box and sum known; the previous tile's flush is done (cells back at zero)
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
        int pitch = ext[2] <= 8 ? 8 : (ext[2] <= 24 ? 24 : (ext[2] <= 40 ? 40 : (ext[2] <= 56 ? 56 : 0)));
        if (given && ((unsigned)ext[0] > 4096u || (unsigned)ext[1] > 4096u))
            pitch = 0;
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const int nbox = nrows * pitch;
        if (a.hint && tid == 0 && half < 0 && (pitch == 0 || nbox > a.small_cap))
            atomicAdd(a.hint, TX / kT);
        bool direct_tile = false;
        if (pitch == 0 || nbox > a.box_cap) {
            if (given && half < 0 && 2 * tx2 + 1 < a.tiles_x) {
                half_next = 0;
                continue;
            }
            direct_tile = true;
        }
        const bool interior = b0[0] >= 0 && b0[0] + ext[0] <= a.in_len[0] && b0[1] >= 0 && b0[1] + ext[1] <= a.in_len[1] &&
                              b0[2] >= 0 && b0[2] + ext[2] <= a.in_len[2];

        for (int ss = 0; ss < nsteps; ++ss, ++phase) {
            float* dst = dx + (STEPS ? steps[2 * ss] : 0);
            float* gsum = reinterpret_cast<float*>(smem + kSOffSum) + (phase & 1) * 4;
            if (ss > 0) {
                if (producer) {
                    float gm = 0.f;
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const bool inb = vyx && oz0 + ZSTEP * i < a.out_z;
                        gval[i] = inb ? dy[steps[2 * ss + 1] + ooff0 + i * ostep] : 0.f;
                        gm += (__float_as_int(gval[i]) & 0x7f800000) == 0x7f800000 ? 0.f : fabsf(gval[i]);
                    }
                    gm = wave_sum(gm);
                    if (lane == 0)
                        gsum[pw] = gm;
                }
                lds_barrier();           // sum known; the previous step's flush is done with the box
            }
            const float gtot = unif((gsum[0] + gsum[1]) + (gsum[2] + gsum[3]));
            constexpr float kC = (float)((2147483648.0 - 1024.0) / ((ORDER == 1 ? 1.0 : ORDER == 2 ? 0.4219 : 0.2963) * 1.001));
            const float scale = gtot > 0.f ? fminf(kC * __frcp_rn(gtot), 3.0e38f) : 0.f;
            const float inv_scale = gtot > 0.f ? __frcp_rn(scale) : 0.f;

            // ---- rounds: producers make the records of voxel r, consumers scatter the records of voxel r - 1 ----
#pragma unroll 1
            for (int r = 0; r <= NV; ++r) {
                if (producer) {
                    if (r < NV) {
                        float gv = gval[0];
#pragma unroll
                        for (int k = 1; k < NV; ++k)
                            gv = r == k ? gval[k] : gv;
                        int st[3] = {0, 0, 0};
                        float fr[3] = {0.f, 0.f, 0.f};
                        int addr = -1;
                        if (gv != 0.f && voxel(r, st, fr)) {
                            const int rz = st[0] - b0[0], ry = st[1] - b0[1], rx = st[2] - b0[2];
                            const bool outside = direct_tile || (given && (rz < 0 || rz + ORDER >= ext[0] || ry < 0 || ry + ORDER >= ext[1] ||
                                                                           rx < 0 || rx + ORDER >= ext[2]));
                            if ((__float_as_int(gv) & 0x7f800000) == 0x7f800000 || outside) {
                                // inf / NaN gradient (no fixed-point scale), or a window outside a stale box: float atomics
                                // straight to global memory, by the producer itself (rare, rolled loop)
                                float w0[NT], w1[NT], w2[NT];
                                weights_from_frac<float, ORDER>(fr[0], w0);
                                weights_from_frac<float, ORDER>(fr[1], w1);
                                weights_from_frac<float, ORDER>(fr[2], w2);
#pragma unroll 1
                                for (int t = 0; t < NT * NT * NT; ++t) {
                                    const int l0 = t / (NT * NT), l1 = (t / NT) % NT, l2 = t % NT;
                                    const int zs = mirror_i32(st[0] + l0, a.in_len[0]);
                                    const int ys = mirror_i32(st[1] + l1, a.in_len[1]);
                                    const int xs = mirror_i32(st[2] + l2, a.in_len[2]);
                                    float wp = w0[0], wq = w1[0], wr = w2[0];
#pragma unroll
                                    for (int l = 1; l < NT; ++l) {
                                        wp = l0 == l ? w0[l] : wp;
                                        wq = l1 == l ? w1[l] : wq;
                                        wr = l2 == l ? w2[l] : wr;
                                    }
                                    unsafeAtomicAdd(dst + (zs * vol_sz + ys * vol_sy + xs), gv * wp * wq * wr);
                                }
                            } else {
                                addr = (rz * by + ry) * pitch + rx;
                            }
                        }
                        int* rec = ring + ((r & 1) * 4 + pw) * 5 * 64 + lane;
                        rec[0] = addr;
                        rec[64] = __float_as_int(fr[0]);
                        rec[128] = __float_as_int(fr[1]);
                        rec[192] = __float_as_int(fr[2]);
                        rec[256] = __float_as_int(gv * scale);
                    }
                } else if (r > 0) {
                    const int* rec = ring + (((r - 1) & 1) * 4 + pw) * 5 * 64 + lane;
                    const int addr = rec[0];
                    if (addr >= 0) {
                        const float f0 = __int_as_float(rec[64]), f1 = __int_as_float(rec[128]), f2 = __int_as_float(rec[192]);
                        const float gs = __int_as_float(rec[256]);
                        float w0[NT], w1[NT], w2[NT];
                        weights_from_frac<float, ORDER>(f0, w0);
                        weights_from_frac<float, ORDER>(f1, w1);
                        weights_from_frac<float, ORDER>(f2, w2);
                        int* bp = box + addr;
#pragma unroll
                        for (int l0 = 0; l0 < NT; ++l0) {
                            const float g0 = gs * w0[l0];
#pragma unroll
                            for (int l1 = 0; l1 < NT; ++l1) {
                                const float g1 = g0 * w1[l1];
                                int* rp = bp + (l0 * by + l1) * pitch;
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
                }
                lds_barrier();           // round r done: records of voxel r are in, records of voxel r - 1 are scattered
            }
            // (the last round's barrier: all contributions are in)
            // flush: half a wave per box row, lanes along x -- one float atomic per touched source element
            {
                constexpr int FL = 32;
                constexpr int FR = kSBlock / FL;
                constexpr int FU = 4;
                const int sub = tid & (FL - 1);
                const int rslot = tid / FL;
                const float inv_by = 1.f / (float)by;
                const int nr = direct_tile ? 0 : nrows;
                const int dz8 = (int)(((float)FR + 0.5f) * inv_by), dy8 = FR - dz8 * by;
                const int step8 = dz8 * vol_sz + dy8 * vol_sy, wrapfix = vol_sz - by * vol_sy;
                for (int xo = 0; xo < ext[2]; xo += FL) {
                    const int xi = xo + sub;
                    const bool xin = xi < ext[2];
                    const int xs = interior ? xi : mirror_i32(b0[2] + xi, a.in_len[2]);
                    int zr = (int)(((float)rslot + 0.5f) * inv_by), yr = rslot - zr * by;
                    int rowoff = (b0[0] + zr) * vol_sz + (b0[1] + yr) * vol_sy + b0[2];
                    for (int r0 = rslot; r0 < nr; r0 += FU * FR) {
                        int acc[FU];
#pragma unroll
                        for (int k = 0; k < FU; ++k) {
                            const int r = r0 + k * FR;
                            acc[k] = (xin && r < nr) ? __hip_atomic_exchange(&box[r * pitch + xi], 0, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_WORKGROUP)
                                                     : 0;
                        }
#pragma unroll
                        for (int k = 0; k < FU; ++k) {
                            if (acc[k] != 0) {
                                int off = rowoff;
                                if (!interior)
                                    off = mirror_i32(b0[0] + zr, a.in_len[0]) * vol_sz + mirror_i32(b0[1] + yr, a.in_len[1]) * vol_sy;
                                unsafeAtomicAdd(dst + (off + xs), (float)acc[k] * inv_scale);
                            }
                            yr += dy8;
                            zr += dz8;
                            rowoff += step8;
                            while (yr >= by) {
                                yr -= by;
                                zr += 1;
                                rowoff += wrapfix;
                            }
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

size_t k2z_lds_bytes(int* box_cap)
{
    *box_cap = kGradBoxBytes / 4;
    return kSOffBox + kGradBoxBytes;
}

// hg: the argument block the launcher fills for hot_grad_kernel (volume / image pointers and strides, geometry, boxes,
// hint, step axes); zg: the z-walk tables in the geometry buffer (R, z table, ZGen, step offsets)
hipError_t launch_k2z(const HotGeom& hg, const ZGeom& zg, int order, size_t lds, hipStream_t stream)
{
    ZGrad a;
    memset(&a, 0, sizeof(a));
    a.dx = hg.vol_w;
    a.dy = hg.img_r;
    a.r = zg.r;
    a.zt = zg.zt;
    a.boxes = hg.use_boxes ? hg.boxes : nullptr;
    a.steps = zg.steps;
    a.hint = hg.hint;
    a.vol_bstride = hg.vol_bstride;
    a.img_bstride = hg.img_bstride;
    a.r_bstride = zg.r_bstride;
    a.vol_sy = hg.vol_sy;
    a.vol_sz = hg.vol_sz;
    a.img_sy = hg.img_sy;
    a.img_sz = hg.img_sz;
    a.box_cap = hg.box_cap;
    a.small_cap = hg.small_cap;
    a.tiles_z = hg.tiles[0];
    a.tiles_y = hg.tiles[1];
    a.tiles_x = hg.tiles[2];
    a.tiles_x2 = (hg.tiles[2] + 1) / 2;
    a.ntiles = hg.ntiles;
    a.strip_tiles = zg.strip_tiles;
    a.nstrips = a.tiles_y * a.tiles_x2 * ((a.tiles_z + a.strip_tiles - 1) / a.strip_tiles);
    const int nb = hg.total_strips / (hg.nstrips > 0 ? hg.nstrips : 1);
    a.total_strips = a.nstrips * nb;
    a.rcol_bytes = 32 * zg.ncpz;
    a.out_z = hg.out_len[0];
    a.out_y = hg.out_len[1];
    a.out_x = hg.out_len[2];
    for (int h = 0; h < 3; ++h) {
        a.in_len[h] = hg.in_len[h];
        a.off[h] = hg.off[h];
    }
    a.nsteps = (int)hg.nsteps;
    a.dbg = hg.dbg;
    a.deal = 4 * a.tiles_x2;
    while (a.deal > 1 && a.total_strips < 16 * a.deal)
        a.deal >>= 1;
    const int chunks = (a.total_strips + a.deal - 1) / a.deal;
    const unsigned nblk = (unsigned)(((chunks + 7) / 8) * 8 * a.deal);
    czgen_p zn = (czgen_p)zg.zgen;
    const bool steps = hg.nstep != 0;
    const bool onewave = ed_env("EDHIP_K2S") == nullptr;       // (EDHIP_K2S: producer / consumer waves; else the four-wave kernel)
    const bool small = ed_env("EDHIP_K2Y") != nullptr;         // (profiling build: 8^3 tiles, two waves per workgroup)
    ZGrad a8 = a;
    unsigned nblk8 = 0;
    size_t lds8 = 0;
    if (small) {
        a8.nstrips = a.tiles_y * a.tiles_x * ((a.tiles_z + a.strip_tiles - 1) / a.strip_tiles);
        a8.total_strips = a8.nstrips * nb;
        a8.box_cap = 4800;
        a8.small_cap = 4800;
        a8.deal = 4 * a.tiles_x;
        while (a8.deal > 1 && a8.total_strips < 16 * a8.deal)
            a8.deal >>= 1;
        const int ch8 = (a8.total_strips + a8.deal - 1) / a8.deal;
        nblk8 = (unsigned)(((ch8 + 7) / 8) * 8 * a8.deal);
        lds8 = kGOffBox + 4800 * 4;
    }
#define ED_K2Z_GO(O, A, S)                                                                                              \
    do {                                                                                                                \
        if (small)                                                                                                      \
            hipLaunchKernelGGL((k2y_grad_kernel<O, A, S>), dim3(nblk8), dim3(128), lds8, stream, a8, zn);               \
        else if (onewave)                                                                                               \
            hipLaunchKernelGGL((k2z_grad_kernel<O, A, S>), dim3(nblk), dim3(kBlock), lds, stream, a, zn);               \
        else                                                                                                            \
            hipLaunchKernelGGL((k2s_grad_kernel<O, A, S>), dim3(nblk), dim3(kSBlock), lds, stream, a, zn);              \
    } while (0)
#define ED_K2Z_ORDER(O)                                                                   \
    do {                                                                                  \
        if (hg.has_affine) { if (steps) ED_K2Z_GO(O, true, true); else ED_K2Z_GO(O, true, false); }   \
        else { if (steps) ED_K2Z_GO(O, false, true); else ED_K2Z_GO(O, false, false); }   \
    } while (0)
    switch (order) {
    case 1: ED_K2Z_ORDER(1); break;
    case 2: ED_K2Z_ORDER(2); break;
    case 3: ED_K2Z_ORDER(3); break;
    default: return hipErrorNotSupported;
    }
#undef ED_K2Z_ORDER
#undef ED_K2Z_GO
    return hipGetLastError();
}

}  // namespace tile
}  // namespace ed
