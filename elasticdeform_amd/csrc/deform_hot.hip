// deform_hot.hip -- K2, the gradient scatter-add of the benchmark case: float32 volumes, 3 deformed axes, unit
// stride along x on both sides, spline orders 1-3 (deform.c:926-997).  Tiles of 8 (z) x 8 (y) x 16 (x) output voxels,
// four voxels per lane; taps are scattered into fixed-point LDS cells with integer atomics and flushed with one float
// atomic per touched source element (see deform_tile.hip for the scale's no-overflow bound).  The forward kernel of
// this case is deform_k1.hip; the kernels both replaced live in experiments/deform_hot_r4.hip (profiling build only).
//
// Built around the instruction costs measured on MI355X (profiles/r02_ubench_valu.txt, r02_ubench_lds.txt):
//   * the kernel argument block is small (HotGeom) and the rarely used uniform values live in LDS;
//   * coordinates: without an affine map floor / fraction are taken of the DISPLACEMENT alone and the output index
//     is added as an integer; the boundary map is one divergent region for the few lanes that leave the array;
//   * Q rows are [control column][component padded to 4]: a voxel's 12 fp64 taps are 4 ds_read_b128 + 4 ds_read_b64;
//   * bounding-box reductions finish inside the wave with DPP row_bcast steps (no v_readlane);
//   * ONE voxel's registers at a time: the box pass computes window starts for the bounding box only and the scatter
//     pass computes them again right before a voxel's 64 atomics (keeping four voxels' state spilled 33-79 VGPRs).
#include "ed_hot.h"

namespace ed {
namespace tile {

namespace {

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
    // (experiment 1 << 21: 16 consecutive x per 16 lanes; with 1 << 22, row pitch 32, the bank of a cell then is its
    // x alone -- tools/sim/conflicts_k2_4wave.py: 1.26 LDS cycles per 16-lane group against 1.73)
    const bool rows16 = TX == 16 && ED_DBG(hg.dbg, 1 << 21);
    const int xx = (TX == 16 && !rows16) ? 2 * (tid & 7) + ((tid >> 4) & 1) : (tid & (TX - 1));
    const int yy = (TX == 16 && !rows16) ? 4 * ((tid >> 6) & 1) + ((tid >> 5) & 1) + 2 * ((tid >> 3) & 1) : ((tid / TX) & 7);
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
            // (canonical empty box: the stored one may hold the reduction's start values, and `start - INT_MAX` wraps)
            b0[0] = b0[1] = b0[2] = 0;
            bhi[0] = bhi[1] = bhi[2] = -1;
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
        if (ED_DBG(hg.dbg, 1 << 22))
            pitch = ext[2] <= 32 ? 32 : 0;
        if (given && ((unsigned)ext[0] > 4096u || (unsigned)ext[1] > 4096u))
            pitch = 0;          // (a handed-over box is not trusted with the products below)
        const int by = ext[1];
        const int nrows = ext[0] * by;
        const int nbox = nrows * pitch;
        if (hg.hint && tid == 0 && half < 0 && (pitch == 0 || nbox > hg.small_cap))
            atomicAdd(hg.hint, (hg.large_cap > 0 && (pitch == 0 || nbox > hg.large_cap)) ? kHintHuge * (TX / kT) : TX / kT);   // spill feedback, in 8-wide tiles
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
                // A lane walks the box rows rslot, rslot + FR, ...: (z, y) of its row and the row's offset in the
                // volume move by the same uniform step every time, with one wrap of y.  (The offset used to be
                // rebuilt per touched row from the row number -- a float division and three 32-bit multiplies, which
                // issue at a quarter of the rate: ~24 issue slots per row where ~6 do.)
                const int dz8 = (int)(((float)FR + 0.5f) * inv_by), dy8 = FR - dz8 * by;
                const int step8 = dz8 * hg.vol_sz + dy8 * hg.vol_sy, wrapfix = hg.vol_sz - by * hg.vol_sy;
                for (int xo = 0; xo < ext[2]; xo += FL) {
                    const int xi = xo + sub;
                    const bool xin = xi < ext[2];
                    const int xs = interior ? xi : mirror_i32(b0[2] + xi, hg.in_len[2]);
                    int zr = (int)(((float)rslot + 0.5f) * inv_by), yr = rslot - zr * by;
                    int rowoff = (b0[0] + zr) * hg.vol_sz + (b0[1] + yr) * hg.vol_sy + b0[2];
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
                                int off = rowoff;
                                if (!interior)
                                    off = mirror_i32(b0[0] + zr, hg.in_len[0]) * hg.vol_sz +
                                          mirror_i32(b0[1] + yr, hg.in_len[1]) * hg.vol_sy;
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

template <int ORDER>
hipError_t launch_grad_order(const HotGeom& hg, unsigned nblk, size_t lds, hipStream_t stream)
{
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
    // (a cell block beyond 64 KiB -- the "huge" boxes of strongly deformed volumes, two workgroups per CU -- needs the
    // kernel's dynamic-LDS limit raised: once per kernel and DEVICE (the attribute belongs to the device's copy of the
    // function); not per call, so that a call captured into a HIP graph makes no attribute call once the geometry has
    // run outside a capture)
    auto go = [&](auto kern, unsigned& raised_on) -> hipError_t {
        if (lds > 64 * 1024) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            const unsigned bit = 1u << (dev & 31);
            if (!(__atomic_load_n(&raised_on, __ATOMIC_RELAXED) & bit)) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess)
                    return e;
                __atomic_fetch_or(&raised_on, bit, __ATOMIC_RELAXED);
            }
        }
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(kBlock), lds, stream, hg);
        return hipGetLastError();
    };
    static unsigned raised[4] = {0, 0, 0, 0};        // (per ORDER: this is a function template)
    if (hg.io16) {
        if constexpr (ORDER <= 3) {
            if (hg.has_affine)
                return go(hot_grad_kernel<ORDER, true, 4, 16, 1, true>, raised[0]);
            return go(hot_grad_kernel<ORDER, false, 4, 16, 1, true>, raised[1]);
        }
        return hipErrorNotSupported;
    }
    if (hg.has_affine)
        return go(hot_grad_kernel<ORDER, true, 4, 16>, raised[2]);
    return go(hot_grad_kernel<ORDER, false, 4, 16>, raised[3]);
}

}  // namespace

#ifdef EDHIP_EXPERIMENTS
hipError_t launch_hot_fwd_r4(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream);   // experiments/deform_hot_r4.hip
#else
// (the records route and the round-4 forward kernel exist in the profiling build only)
hipError_t launch_hot_records(const HotGeom&, int, unsigned, size_t, hipStream_t) { return hipErrorNotSupported; }
size_t hot_grad2_lds_bytes(int* box_cap, bool) { *box_cap = 0; return 0; }
hipError_t launch_hot_grad2(const HotGeom&, int, unsigned, size_t, hipStream_t) { return hipErrorNotSupported; }
#endif

// LDS: x table | reduction slots | wave sums | parameters | 64 Q rows | box.  Returns 0 when the
// control grid is too wide for a useful box (the general kernels take the call).
size_t hot_lds_bytes(bool gradient, int ncpx, int* box_cap, int* off_box, int level)
{
    const size_t q = (size_t)kT * kT * 32 * (size_t)ncpx;
    const size_t off = (kOffQ + q + 15) & ~(size_t)15;
    *off_box = (int)off;
    // Large boxes (three workgroups per CU instead of four) for strongly deformed volumes.  256^3 order 3,
    // whole call, standard -> large: sigma 5 forward 262 -> 286 us, gradient 325 -> 357; sigma 10 forward
    // 368 -> 340, gradient 501 -> 427; sigma 15 forward 614 -> 496, gradient 1412 -> 1264
    // (profiles/r03_bench_misc.txt): launch_tile picks them when recent calls of the geometry spilled.
    const bool large = level >= 1;
    if (gradient) {
        // Huge boxes (64 KiB of cells, two workgroups per CU) where a tenth of the tiles does not fit the large ones.
        // 256^3 order 3, gradient call after a forward call, large -> huge: sigma 10 434 -> 497 us, sigma 12.5
        // 546 -> 528, sigma 15 837 -> 579, sigma 20 3600 -> 779 (profiles/r06_k2_box_sweep.txt)
        size_t box = level >= 2 ? 64 * 1024 : (large ? 36 * 1024 : kGradBoxBytes);
        if (const char* kb = ed_env("EDHIP_GRAD_BOX_KB"))
            box = (size_t)atoi(kb) * 1024;
        *box_cap = (int)(box / 4);
        const size_t total = off + box;
        return total <= 80 * 1024 ? total : 0;
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
    if (!gradient) {
#ifdef EDHIP_EXPERIMENTS
        return launch_hot_fwd_r4(hg, order, nblk, lds, stream);
#else
        return hipErrorNotSupported;      // (the forward kernel of this case is deform_k1.hip)
#endif
    }
    switch (order) {
    case 1: return launch_grad_order<1>(hg, nblk, lds, stream);
    case 2: return launch_grad_order<2>(hg, nblk, lds, stream);
    case 3: return launch_grad_order<3>(hg, nblk, lds, stream);
    case 4: return launch_grad_order<4>(hg, nblk, lds, stream);
    case 5: return launch_grad_order<5>(hg, nblk, lds, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace tile
}  // namespace ed
