// experiments/deform_hot_r4.hip -- PROFILING BUILD ONLY (make EXPERIMENTS=1): the round-4 kernels that deform_k1.hip
// (K1) replaced, kept for A/B measurements on one box (EDHIP_K1_OLD=1), with their ablation ladder (EDHIP_HOT_ABL),
// and the records route of round 4 (K1 leaves per-voxel coordinate records, hot_grad2_kernel scatters merged
// x-neighbour pairs from compacted work lists: EDHIP_RECORDS=1; measured, not faster -- profiles/r04_records_route.txt).
// Nothing in here is part of the shipped library.
//
// K1 of round 4: same algorithm and LDS tiling as the general kernels of deform_tile.hip, the exact bounding box of a
// tile's tap windows reduced per tile (DPP + LDS atomics), coordinates of tile t + 1 under the LDS-DMA copies of tile t.
#include "ed_hot.h"

#ifndef EDHIP_EXPERIMENTS
#error "experiments/deform_hot_r4.hip is part of the profiling build only"
#endif

namespace ed {
namespace tile {

namespace {

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
hipError_t launch_fwd_order(const HotGeom& hg, unsigned nblk, size_t lds, hipStream_t stream)
{
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
    if (hg.has_affine)
        hipLaunchKernelGGL((hot_fwd_kernel<ORDER, true>), dim3(nblk), dim3(kBlock), lds, stream, hg);
    else
        hipLaunchKernelGGL((hot_fwd_kernel<ORDER, false>), dim3(nblk), dim3(kBlock), lds, stream, hg);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_hot_fwd_r4(const HotGeom& hg, int order, unsigned nblk, size_t lds, hipStream_t stream)
{
    switch (order) {
    case 1: return launch_fwd_order<1>(hg, nblk, lds, stream);
    case 2: return launch_fwd_order<2>(hg, nblk, lds, stream);
    case 3: return launch_fwd_order<3>(hg, nblk, lds, stream);
    case 4: return launch_fwd_order<4>(hg, nblk, lds, stream);
    case 5: return launch_fwd_order<5>(hg, nblk, lds, stream);
    default: return hipErrorNotSupported;
    }
}

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

}  // namespace tile
}  // namespace ed
