// deform_exact.hip -- K1/K2 "exact" kernels: the reference's per-voxel pipeline in fp64, in the
// reference's own evaluation order, one thread per (output voxel, step).
//
// Compiled with -ffp-contract=off: every product and sum below rounds exactly like the x86-64
// build of /root/reference/elasticdeform/deform.c (no FMA there), so float64 outputs, integer
// label volumes and order-0 resampling are bit-comparable with the reference.  Used for every
// dtype other than float32 by default, and for float32 when the caller asks (EDHIP_FLAG_EXACT).
//
// Restates DeformGrid's hot loop, deform.c:649-1001:
//   displacement B-spline   :650-758      coordinate + boundary map   :768-824
//   forward tap gather      :841-924      gradient scatter-add        :926-997
// The scatter-add uses device-scope atomics (the reference is sequential); integer accumulation
// is exact under reordering, floating-point accumulation differs from the reference only by the
// order of the additions.
#include "ed_device.h"
#include "ed_exact_coord.h"
#include "ed_params.h"

namespace ed {

namespace {

// *(T*)p += (T)t for every dtype -- deform.c:309-312,974-987
__device__ __forceinline__ void atomic_accumulate(char* p, int dt, double t)
{
    switch (dt) {
    case EDHIP_F64: unsafeAtomicAdd((double*)p, t); break;
    case EDHIP_F32: unsafeAtomicAdd((float*)p, (float)t); break;
    case EDHIP_I32: atomicAdd((unsigned int*)p, (unsigned int)trunc_i32(t)); break;
    case EDHIP_U32: atomicAdd((unsigned int*)p, (unsigned int)trunc_i64(t)); break;
    case EDHIP_I64: atomicAdd((unsigned long long*)p, (unsigned long long)trunc_i64(t)); break;
    case EDHIP_U64: atomicAdd((unsigned long long*)p, (unsigned long long)trunc_u64(t)); break;
    case EDHIP_F16:
    case EDHIP_BF16: {
        // 16-bit float lane of a 32-bit word: *(T*)p += (T)t, each add rounded to the storage type
        const uintptr_t a = (uintptr_t)p;
        unsigned int* word = (unsigned int*)(a & ~(uintptr_t)3);
        const unsigned int shift = (unsigned int)(a & 2) * 8;
        uint16_t add_bits;
        store_cast((char*)&add_bits, dt, t);
        const double addend = load_as_double((const char*)&add_bits, dt);
        if (addend == 0.0)
            break;
        unsigned int old = *word, assumed;
        do {
            assumed = old;
            const uint16_t cur = (uint16_t)(assumed >> shift);
            uint16_t upd;
            store_cast((char*)&upd, dt, load_as_double((const char*)&cur, dt) + addend);
            const unsigned int next = (assumed & ~(0xffffu << shift)) | ((unsigned int)upd << shift);
            old = atomicCAS(word, assumed, next);
        } while (old != assumed);
        break;
    }
    default: {
        // 8- and 16-bit lanes of a 32-bit word: wrap-around add inside the lane via CAS
        const int bytes = (dt == EDHIP_U16 || dt == EDHIP_I16) ? 2 : 1;
        const unsigned int lane_mask = bytes == 2 ? 0xffffu : 0xffu;
        const unsigned int addend = (unsigned int)trunc_i32(t) & lane_mask;
        if (addend == 0)
            break;
        const uintptr_t a = (uintptr_t)p;
        unsigned int* word = (unsigned int*)(a & ~(uintptr_t)3);
        const unsigned int shift = (unsigned int)(a & 3) * 8;
        unsigned int old = *word, assumed;
        do {
            assumed = old;
            const unsigned int cur = (assumed >> shift) & lane_mask;
            const unsigned int upd = (cur + addend) & lane_mask;
            const unsigned int next = (assumed & ~(lane_mask << shift)) | (upd << shift);
            old = atomicCAS(word, assumed, next);
        } while (old != assumed);
        break;
    }
    }
}

// edhip_source_box: floor(min) / ceil(max) of the raw source coordinates over all output voxels
__global__ void source_box_init_kernel(int* box, int naxis)
{
    if ((int)threadIdx.x < 2 * naxis)
        box[threadIdx.x] = (threadIdx.x & 1) ? (int)0x80000000 : 0x7fffffff;
}

// (order + 1)^3 taps of one voxel, lexicographic, summed exactly like the digit-counter loop of
// deform_exact_kernel (value, then * w0, * w1, * w2, then added: deform.c:863-899), with the loads of
// a row issued together.  S: a type whose conversion to double is the plain C one.
template <typename S>
__device__ __forceinline__ double exact_rows3(const char* base, const int64_t (&tap)[3][6], const double (&w)[3][6],
                                              int order)
{
#pragma clang fp contract(off)
    double t = 0.0;
    for (int l0 = 0; l0 <= order; ++l0) {
        for (int l1 = 0; l1 <= order; l1 += 2) {      // two rows (up to 12 loads) in flight
            const bool two = l1 + 1 <= order;
            const char* row0 = base + (tap[0][l0] + tap[1][l1]);
            const char* row1 = base + (tap[0][l0] + tap[1][two ? l1 + 1 : l1]);
            S v0[6], v1[6];
#pragma unroll
            for (int l2 = 0; l2 < 6; ++l2) {    // (taps past the order re-read tap 0: no branch around a load)
                v0[l2] = *reinterpret_cast<const S*>(row0 + tap[2][l2 <= order ? l2 : 0]);
                v1[l2] = *reinterpret_cast<const S*>(row1 + tap[2][l2 <= order ? l2 : 0]);
            }
#pragma unroll
            for (int l2 = 0; l2 < 6; ++l2) {
                if (l2 <= order) {
                    double coeff = (double)v0[l2];
                    if (order > 0) {
                        coeff *= w[0][l0];
                        coeff *= w[1][l1];
                        coeff *= w[2][l2];
                    }
                    t += coeff;
                }
            }
            if (two) {
#pragma unroll
                for (int l2 = 0; l2 < 6; ++l2) {
                    if (l2 <= order) {
                        double coeff = (double)v1[l2];
                        if (order > 0) {
                            coeff *= w[0][l0];
                            coeff *= w[1][l1 + 1];
                            coeff *= w[2][l2];
                        }
                        t += coeff;
                    }
                }
            }
        }
    }
    return t;
}

// The control grid is first copied into LDS as doubles (component-major, C order): at 64 x naxis
// dependent global loads per voxel the kernel was latency-bound (326 us for a 64^3 crop).
template <int NAXIS>
__global__ __launch_bounds__(256) void source_box_kernel(const GridGeom g, int* box)
{
    extern __shared__ double sgrid[];       // [NAXIS][ncp_0]...[ncp_{NAXIS-1}]
    const int per = stage_grid_lds<NAXIS>(g, sgrid);
    __syncthreads();

    int lo[NAXIS], hi[NAXIS];
#pragma unroll
    for (int h = 0; h < NAXIS; ++h) {
        lo[h] = 0x7fffffff;
        hi[h] = (int)0x80000000;
    }
    for (int64_t kk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; kk < g.nvox;
         kk += (int64_t)gridDim.x * blockDim.x) {
        int64_t o[NAXIS];
        int64_t r = kk;
#pragma unroll
        for (int k = NAXIS - 1; k >= 0; --k) {
            const int64_t q = r / g.out_len[k];
            o[k] = r - q * g.out_len[k];
            r = q;
        }
        double displ[NAXIS];
        eval_displacement_lds<NAXIS>(g, sgrid, per, o, displ);
#pragma unroll
        for (int h = 0; h < NAXIS; ++h) {
            const double acc = displ[h];
            double c = raw_coordinate<NAXIS>(g, o, h, acc);
            if (!(c == c))
                c = 0.0;                                        // NaN grid entries: no constraint
            c = c < -1e9 ? -1e9 : (c > 1e9 ? 1e9 : c);
            // (callers keep a sample of slack around the box)
            lo[h] = min(lo[h], (int)floor(c));
            hi[h] = max(hi[h], (int)ceil(c));
        }
    }
#pragma unroll
    for (int h = 0; h < NAXIS; ++h) {
        for (int m = 32; m >= 1; m >>= 1) {
            lo[h] = min(lo[h], __shfl_xor(lo[h], m));
            hi[h] = max(hi[h], __shfl_xor(hi[h], m));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&box[2 * h], lo[h]);
            atomicMax(&box[2 * h + 1], hi[h]);
        }
    }
}

// one (output voxel kk, step ss) pair in the reference's evaluation order
template <int NAXIS, bool LDSGRID>
__device__ __forceinline__ void exact_item(const GridGeom& g, const IOView& v, const int gradient,
                                           const double* sgrid, const int per, const int64_t kk, const int64_t ss)
{

    // output voxel index, last deformed axis fastest (from_scipy.h:67-79)
    int64_t o[NAXIS];
    {
        int64_t r = kk;
#pragma unroll
        for (int k = NAXIS - 1; k >= 0; --k) {
            const int64_t q = r / g.out_len[k];
            o[k] = r - q * g.out_len[k];
            r = q;
        }
    }

    double displ[NAXIS];
    if (LDSGRID)
        eval_displacement_lds<NAXIS>(g, sgrid, per, o, displ);
    else
        eval_displacement<NAXIS>(g, o, displ);

    // ---- source coordinate, boundary map, window + weights, deform.c:768-824 ------------------
    const int order = v.order;
    double w[NAXIS][6];
    int64_t tap[NAXIS][6];    // byte offsets of the taps on each deformed input axis
    bool constant = false;
#pragma unroll
    for (int h = 0; h < NAXIS; ++h) {
        const double cc = map_coordinate(raw_coordinate<NAXIS>(g, o, h, displ[h]), g.in_len[h], v.mode);
        if (!constant && cc > -1.0) {
            const int64_t start = window_start(cc, order);
            const bool edge = start < 0 || start + order >= g.in_len[h];
            for (int l = 0; l <= order; ++l) {
                const int64_t idx = edge ? mirror_index(start + l, g.in_len[h]) : start + l;
                tap[h][l] = idx * v.in_stride[h];
            }
            spline_weights(cc, order, w[h]);
        } else {
            constant = true;   // deform.c:819-822 (first failing axis decides; later ones unused)
        }
    }

    // ---- step (non-deformed axes) offsets, first step axis fastest, deform.c:828-838 ----------
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
#pragma unroll
    for (int k = 0; k < NAXIS; ++k)
        out_off += v.out_stride[k] * o[k];
    char* po = v.out + out_off;

    // number of taps (order+1)^NAXIS, walked lexicographically with a digit counter
    int cnt[NAXIS];
    if (!gradient) {
        double t = 0.0;
        bool rows_done = false;
        if (!constant && NAXIS == 3) {
            // three deformed axes, element types with a plain C conversion: the same sum in the same
            // order, but a row of taps (last axis) is loaded before it is accumulated -- the digit
            // counter below issues one dependent global load per tap (64 round trips per voxel at
            // order 3: the exact kernel's time)
            if constexpr (NAXIS == 3) {
                rows_done = true;
                const char* base = v.in + in_off;
                switch (v.in_dtype) {
                case EDHIP_BOOL:
                case EDHIP_U8: t = exact_rows3<uint8_t>(base, tap, w, order); break;
                case EDHIP_I8: t = exact_rows3<int8_t>(base, tap, w, order); break;
                case EDHIP_U16: t = exact_rows3<uint16_t>(base, tap, w, order); break;
                case EDHIP_I16: t = exact_rows3<int16_t>(base, tap, w, order); break;
                case EDHIP_U32: t = exact_rows3<uint32_t>(base, tap, w, order); break;
                case EDHIP_I32: t = exact_rows3<int32_t>(base, tap, w, order); break;
                case EDHIP_U64: t = exact_rows3<uint64_t>(base, tap, w, order); break;
                case EDHIP_I64: t = exact_rows3<int64_t>(base, tap, w, order); break;
                case EDHIP_F32: t = exact_rows3<float>(base, tap, w, order); break;
                case EDHIP_F64: t = exact_rows3<double>(base, tap, w, order); break;
                default: rows_done = false; break;       // 16-bit floats: below
                }
            }
        }
        if (rows_done) {
        } else if (!constant) {                            // deform.c:843-901
#pragma unroll
            for (int k = 0; k < NAXIS; ++k)
                cnt[k] = 0;
            for (;;) {
                int64_t offs = in_off;
#pragma unroll
                for (int k = 0; k < NAXIS; ++k)
                    offs += tap[k][cnt[k]];
                double coeff = load_as_double(v.in + offs, v.in_dtype);
                if (order > 0) {
#pragma unroll
                    for (int k = 0; k < NAXIS; ++k)
                        coeff *= w[k][cnt[k]];
                }
                t += coeff;
                int k = NAXIS - 1;
                for (; k >= 0; --k) {
                    if (cnt[k] < order) {
                        cnt[k]++;
                        break;
                    }
                    cnt[k] = 0;
                }
                if (k < 0)
                    break;
            }
        } else {
            t = v.cval;                                    // deform.c:903
        }
        store_forward(po, v.out_dtype, t);
    } else if (!constant) {                                // deform.c:926-996
        const double grad = load_as_double(po, v.out_dtype);
#pragma unroll
        for (int k = 0; k < NAXIS; ++k)
            cnt[k] = 0;
        for (;;) {
            double coeff = grad;
            if (order > 0) {
#pragma unroll
                for (int k = 0; k < NAXIS; ++k)
                    coeff *= w[k][cnt[k]];
            }
            int64_t offs = in_off;
#pragma unroll
            for (int k = 0; k < NAXIS; ++k)
                offs += tap[k][cnt[k]];
            atomic_accumulate(const_cast<char*>(v.in) + offs, v.in_dtype, coeff);
            int k = NAXIS - 1;
            for (; k >= 0; --k) {
                if (cnt[k] < order) {
                    cnt[k]++;
                    break;
                }
                cnt[k] = 0;
            }
            if (k < 0)
                break;
        }
    }
}

template <int NAXIS, bool LDSGRID>
__global__ __launch_bounds__(256) void deform_exact_kernel(const GridGeom g, const IOView v,
                                                           const int gradient)
{
    extern __shared__ double sgrid[];       // LDSGRID: the control grid as doubles
    int per = 0;
    if (LDSGRID) {
        per = stage_grid_lds<NAXIS>(g, sgrid);
        __syncthreads();
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = g.nvox * v.nsteps;
    if (tid >= total)
        return;
    int64_t kk, ss;
    if (v.steps_fastest) {
        kk = tid / v.nsteps;
        ss = tid - kk * v.nsteps;
    } else {
        ss = tid / g.nvox;
        kk = tid - ss * g.nvox;
    }
    exact_item<NAXIS, LDSGRID>(g, v, gradient, sgrid, per, kk, ss);
}

// The voxels of a list (the near-tie voxels of the integer fast path, deform_wave.hip), forward, every
// step of each: [0] = count, [1 .. cap] = linear output voxel ids.  A list that overflowed its capacity
// (adversarial inputs) means every voxel -- correct, at the exact kernel's speed.
template <bool LDSGRID>
__global__ __launch_bounds__(256) void deform_exact_list_kernel(const GridGeom g, const IOView v,
                                                                const int* list, const int cap)
{
    extern __shared__ double sgrid[];
    int per = 0;
    const int count = list[0];
    if (count == 0)
        return;
    if (LDSGRID) {
        per = stage_grid_lds<3>(g, sgrid);
        __syncthreads();
    }
    const bool all = count > cap;
    const int64_t nv = all ? g.nvox : (int64_t)count;
    const int64_t n = nv * v.nsteps;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iv = e / v.nsteps, ss = e - iv * v.nsteps;
        const int64_t kk = all ? iv : (int64_t)list[1 + iv];
        exact_item<3, LDSGRID>(g, v, 0, sgrid, per, kk, ss);
    }
}

}  // namespace

hipError_t launch_deform_exact_list(const GridGeom& g, const IOView& v, const int* list, int cap,
                                    hipStream_t stream)
{
    if (g.naxis != 3 || g.nvox <= 0)
        return hipErrorInvalidValue;
    int64_t points = 3;
    for (int k = 0; k < 3; ++k)
        points *= g.ncp[k];
    const bool lds_grid = points <= 7680;
    const size_t lds = lds_grid ? (size_t)points * sizeof(double) : 0;
    if (lds_grid)
        hipLaunchKernelGGL((deform_exact_list_kernel<true>), dim3(1024), dim3(256), lds, stream, g, v, list, cap);
    else
        hipLaunchKernelGGL((deform_exact_list_kernel<false>), dim3(1024), dim3(256), 0, stream, g, v, list, cap);
    return hipGetLastError();
}

hipError_t launch_deform_exact(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream)
{
    const int64_t total = g.nvox * v.nsteps;
    if (total <= 0)
        return hipSuccess;
    const int block = 256;
    const int64_t nblk = (total + block - 1) / block;
    if (nblk > 0x7fffffffLL)
        return hipErrorInvalidValue;
    const dim3 grid((unsigned)nblk);
    // small control grids (the normal case) are staged in LDS as doubles
    int64_t points = g.naxis;
    for (int k = 0; k < g.naxis; ++k)
        points *= g.ncp[k];
    const bool lds_grid = points <= 7680;
    const size_t lds = lds_grid ? (size_t)points * sizeof(double) : 0;
#define EDHIP_EXACT(N)                                                                                \
    do {                                                                                              \
        if (lds_grid)                                                                                 \
            hipLaunchKernelGGL((deform_exact_kernel<N, true>), grid, dim3(block), lds, stream, g, v,  \
                               gradient);                                                             \
        else                                                                                          \
            hipLaunchKernelGGL((deform_exact_kernel<N, false>), grid, dim3(block), 0, stream, g, v,   \
                               gradient);                                                             \
    } while (0)
    switch (g.naxis) {
    case 1: EDHIP_EXACT(1); break;
    case 2: EDHIP_EXACT(2); break;
    case 3: EDHIP_EXACT(3); break;
    case 4: EDHIP_EXACT(4); break;
    // five to seven deformed axes: the same kernel (digit counters over the (order + 1)^naxis taps);
    // rare shapes, no tuning -- they exist so that every array the C ABI can describe is served
    case 5: EDHIP_EXACT(5); break;
    case 6: EDHIP_EXACT(6); break;
    case 7: EDHIP_EXACT(7); break;
    default: return hipErrorInvalidValue;
    }
#undef EDHIP_EXACT
    return hipGetLastError();
}

// The conservative form: a B-spline is a convex combination of its coefficients (the cubic basis is
// non-negative and sums to one), so over the output box the displacement component h stays inside
// [min, max] of the (prefiltered) control points whose basis functions reach the box, and the affine part
// is extremal at the box's corners.  One workgroup walks the control points -- O(grid) instead of the
// O(voxels) of source_box_kernel (933 us for a 256^3 output): floor / ceil of that hull, a superset of
// the exact box (a few samples wider for the grids elastic deformation uses).
// mirror index map of deform.c:668-686 in 32-bit arithmetic (control grids are small; a 64-bit modulo costs
// about a microsecond here, and the level-0 load did one per axis and point)
__device__ __forceinline__ int mirror_small(int i, int n)
{
    if (n <= 1)
        return 0;
    const int period = 2 * n - 2;
    if (i < 0)
        i = -i;
    i %= period;
    return i >= n ? period - i : i;
}

constexpr int kHullCap = 3900;          // doubles per refinement buffer (two of them + the partial results: under 64 KiB of static LDS)

__global__ __launch_bounds__(1024) void source_box_hull_kernel(const GridGeom g, int* box, const SourceWindow sw)
{
    // one workgroup per component (= per source axis): 52 us for the three of a 3-D grid in one workgroup
    __shared__ double smin[kMaxAxes][16], smax[kMaxAxes][16];       // (up to 16 waves: the passes are short and latency-bound)
    const int tid = threadIdx.x, wave = tid >> 6;
    const int naxis = g.naxis;
    // control-point index range per axis whose basis functions reach the output box
    int64_t lo_i[kMaxAxes], n_i[kMaxAxes], total = 1;
    for (int k = 0; k < naxis; ++k) {
        const double c0 = control_coordinate(g.ncp[k], g.off[k], g.in_len[k]);
        const double c1 = control_coordinate(g.ncp[k], g.out_len[k] - 1 + g.off[k], g.in_len[k]);
        int64_t a = (int64_t)floor(c0 < c1 ? c0 : c1) - 1, b = (int64_t)floor(c0 < c1 ? c1 : c0) + 2;
        // indices beyond the grid mirror back into it: widen the range by what sticks out, then clip
        if (a < 0) {
            b = b > -a ? b : -a;
            a = 0;
        }
        if (b > g.ncp[k] - 1) {
            const int64_t over = b - (g.ncp[k] - 1);
            a = a < g.ncp[k] - 1 - over ? a : g.ncp[k] - 1 - over;
            b = g.ncp[k] - 1;
        }
        a = a < 0 ? 0 : a;
        if (!(c0 == c0) || !(c1 == c1)) {      // (cannot happen for in_len >= 2; be safe)
            a = 0;
            b = g.ncp[k] - 1;
        }
        lo_i[k] = a;
        n_i[k] = b - a + 1;
        total *= n_i[k];
    }
    // Subdivision (Lane-Riesenfeld, cubic): every level halves the knot spacing -- new points
    // (c[j] + 6 c[j+1] + c[j+2]) / 8 and (c[j+1] + c[j+2]) / 2 -- and the spline stays the same, so the hull of
    // the refined points that reach the output box is still a superset, four times closer per level.  For a
    // random 5^3 grid the raw hull is 9x wider than the displacement's true range over a half-size crop, after
    // two levels 1.2x (profiles/r03_design_sims.txt).  Between levels every axis is trimmed to the points
    // within three spacings of the box, so the refined block stays small; up to two levels, as many as fit
    // the two 32 KiB buffers (large crops of fine grids: fewer levels, then the raw hull below).
    __shared__ double sbuf[2][kHullCap];
    double c0[kMaxAxes], c1[kMaxAxes];
    for (int k = 0; k < naxis; ++k) {
        const double a = control_coordinate(g.ncp[k], g.off[k], g.in_len[k]);
        const double b = control_coordinate(g.ncp[k], g.out_len[k] - 1 + g.off[k], g.in_len[k]);
        c0[k] = a < b ? a : b;
        c1[k] = a < b ? b : a;
    }
    int levels = -1;
    for (int L = 2; L >= 0 && levels < 0; --L) {
        double t = 1.0;
        bool ok = naxis <= 4;
        for (int k = 0; k < naxis; ++k) {
            if (!(c0[k] == c0[k]) || !(c1[k] == c1[k]) || c1[k] - c0[k] > 1e6)
                ok = false;
            else
                t *= ceil((c1[k] - c0[k]) * (double)(1 << L)) + 9.0;      // bound of an axis' points at any step
        }
        if (ok && t <= (double)kHullCap)
            levels = L;
    }
    for (int h = blockIdx.x; h < naxis; h += gridDim.x) {
        double mn = 1e300, mx = -1e300;
        if (levels >= 0) {
            int n[4] = {1, 1, 1, 1};
            double T[4] = {0, 0, 0, 0}, hs[4] = {1, 1, 1, 1};
            int tot = 1;
            for (int k = 0; k < naxis; ++k) {
                const int lo = (int)floor(c0[k]) - 3;
                n[k] = (int)floor(c1[k]) + 4 - lo + 1;
                T[k] = (double)lo;
                tot *= n[k];
            }
            int cur = 0;
            for (int e = tid; e < tot; e += (int)blockDim.x) {          // level 0: the mirror-extended coefficients
                int r = e;
                int64_t off = g.disp_stride[0] * h;
                for (int k = naxis - 1; k >= 0; --k) {
                    const int q = r / n[k];
                    off += (int64_t)mirror_small((int)T[k] + (r - q * n[k]), (int)g.ncp[k]) * g.disp_stride[k + 1];
                    r = q;
                }
                sbuf[0][e] = load_as_double(g.disp + off, g.disp_dtype);
            }
            __syncthreads();
            for (int l = 0; l < levels; ++l) {
                for (int ax = 0; ax < naxis; ++ax) {
                    const int na = n[ax];
                    const double Tn = T[ax] + hs[ax], hn = 0.5 * hs[ax];
                    const int nb = 2 * na - 5;
                    int ia = (int)floor((c0[ax] - 3.0 * hn - Tn) / hn), ib = (int)ceil((c1[ax] + 3.0 * hn - Tn) / hn);
                    ia = ia < 0 ? 0 : ia;
                    ib = ib > nb - 1 ? nb - 1 : ib;
                    const int nn = ib - ia + 1;
                    int sA = 1;                              // stride of `ax` in the source block (last axis fastest)
                    for (int k = naxis - 1; k > ax; --k)
                        sA *= n[k];
                    int inner = sA, tot_out = 1;
                    for (int k = 0; k < naxis; ++k)
                        tot_out *= k == ax ? nn : n[k];
                    const double* A = sbuf[cur];
                    double* B = sbuf[cur ^ 1];
                    for (int e = tid; e < tot_out; e += (int)blockDim.x) {
                        const int in_idx = e % inner;                    // axes after ax
                        const int r = e / inner;
                        const int m = ia + r % nn;                       // refined index along ax
                        const int outer = r / nn;                        // axes before ax
                        const int j = m >> 1;
                        const double* a = A + ((size_t)outer * na + j) * inner + in_idx;
                        B[e] = (m & 1) ? 0.5 * (a[sA] + a[2 * sA]) : 0.125 * (a[0] + 6.0 * a[sA] + a[2 * sA]);
                    }
                    __syncthreads();
                    cur ^= 1;
                    n[ax] = nn;
                    T[ax] = Tn + ia * hn;
                    hs[ax] = hn;
                    tot = tot_out;
                }
            }
            for (int e = tid; e < tot; e += (int)blockDim.x) {
                const double v = sbuf[cur][e];
                if (v == v) {                   // NaN grid entries: no constraint (like the exact kernel)
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
            }
            __syncthreads();                    // the next component reuses the buffers
        } else
        for (int64_t e = tid; e < total; e += (int)blockDim.x) {
            int64_t r = e, off = g.disp_stride[0] * h;
            for (int k = naxis - 1; k >= 0; --k) {
                const int64_t q = r / n_i[k];
                off += (lo_i[k] + (r - q * n_i[k])) * g.disp_stride[k + 1];
                r = q;
            }
            const double v = load_as_double(g.disp + off, g.disp_dtype);
            if (v == v) {                       // NaN grid entries: no constraint (like the exact kernel)
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
        }
        for (int m = 32; m >= 1; m >>= 1) {
            const double a = __shfl_xor(mn, m), b = __shfl_xor(mx, m);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if ((tid & 63) == 0) {
            smin[h][wave] = mn;
            smax[h][wave] = mx;
        }
    }
    __syncthreads();
    if (tid < naxis && (tid % (int)gridDim.x) == (int)blockIdx.x) {
        const int h = tid;
        double mn = smin[h][0], mx = smax[h][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            mn = smin[h][w] < mn ? smin[h][w] : mn;
            mx = smax[h][w] > mx ? smax[h][w] : mx;
        }
        if (mn > mx)
            mn = mx = 0.0;                      // every entry NaN
        double blo, bhi;
        if (g.has_affine) {
            blo = bhi = g.affine[h * (naxis + 1) + naxis];
            for (int l = 0; l < naxis; ++l) {
                const double t = g.affine[h * (naxis + 1) + l] * (double)(g.out_len[l] - 1);
                blo += t < 0.0 ? t : 0.0;
                bhi += t > 0.0 ? t : 0.0;
            }
        } else {
            blo = 0.0;
            bhi = (double)(g.out_len[h] - 1);
        }
        double lo = blo + (double)g.off[h] + mn, hi = bhi + (double)g.off[h] + mx;
        lo = !(lo == lo) ? -1e9 : (lo < -1e9 ? -1e9 : (lo > 1e9 ? 1e9 : lo));
        hi = !(hi == hi) ? 1e9 : (hi < -1e9 ? -1e9 : (hi > 1e9 ? 1e9 : hi));
        const int blo_i = (int)floor(lo) - 1;   // (one sample for the rounding of the sums above)
        const int bhi_i = (int)ceil(hi) + 1;
        box[2 * h] = blo_i;
        box[2 * h + 1] = bhi_i;
        if (sw.out) {
            // The filter window of this source axis, left on the device (edhip_source_window): the tap window
            // of deform.c:783-813 around the coordinate range, the boundary map's say where the range leaves
            // the array ('nearest' / 'constant' clip it, the folding modes take the whole axis), the filter's
            // decay margin, rows of the last axis on vector boundaries, at least `minlen` samples.
            const int n = sw.shape[sw.axis[h]];
            int wl = blo_i - sw.order / 2 - 1, wh = bhi_i + sw.order - sw.order / 2 + 1;      // inclusive
            if (wl < 0 || wh > n - 1) {
                if (sw.mode == EDHIP_MODE_NEAREST || sw.mode == EDHIP_MODE_CONSTANT) {
                    const bool cl = wl < 0, ch = wh > n - 1;
                    wl = wl < 0 ? 0 : (wl > n - 1 ? n - 1 : wl);
                    wh = wh > n - 1 ? n - 1 : (wh < 0 ? 0 : wh);
                    // windows that stick out are mirror-indexed (deform.c:795-813): up to `order` samples inward
                    if (cl)
                        wh = wh > (sw.order < n - 1 ? sw.order : n - 1) ? wh : (sw.order < n - 1 ? sw.order : n - 1);
                    if (ch)
                        wl = wl < (n - 1 - sw.order > 0 ? n - 1 - sw.order : 0) ? wl : (n - 1 - sw.order > 0 ? n - 1 - sw.order : 0);
                } else {
                    wl = 0;
                    wh = n - 1;
                }
            }
            int w0 = wl - sw.margin, w1 = wh + 1 + sw.margin;
            w0 = w0 < 0 ? 0 : w0;
            w1 = w1 > n ? n : w1;
            if (w1 - w0 < sw.minlen) {
                w0 = w0 < n - sw.minlen ? w0 : n - sw.minlen;
                w0 = w0 < 0 ? 0 : w0;
                w1 = w0 + sw.minlen < n ? w0 + sw.minlen : n;
            }
            if (sw.axis[h] == sw.ndim - 1 && sw.align > 1) {
                w0 -= w0 % sw.align;
                w1 = (w1 + sw.align - 1) / sw.align * sw.align;
                w1 = w1 > n ? n : w1;
            }
            sw.out[2 * sw.axis[h]] = w0;
            sw.out[2 * sw.axis[h] + 1] = w1;
        }
    }
    if (sw.out && blockIdx.x == 0 && tid < sw.ndim) {
        bool deformed = false;
        for (int k = 0; k < naxis; ++k)
            deformed = deformed || sw.axis[k] == tid;
        if (!deformed) {
            sw.out[2 * tid] = 0;
            sw.out[2 * tid + 1] = sw.shape[tid];
        }
    }
}

hipError_t launch_source_box(const GridGeom& g, int* box, hipStream_t stream, bool conservative, const SourceWindow* sw)
{
    if ((conservative || sw) && g.nvox > 0) {
        const SourceWindow none{};
        hipLaunchKernelGGL(source_box_hull_kernel, dim3((unsigned)g.naxis), dim3(1024), 0, stream, g, box, sw ? *sw : none);
        return hipGetLastError();
    }
    if (sw)
        return hipErrorNotSupported;
    hipLaunchKernelGGL(source_box_init_kernel, dim3(1), dim3(64), 0, stream, box, g.naxis);
    if (g.nvox <= 0)
        return hipGetLastError();
    int64_t points = g.naxis;
    for (int k = 0; k < g.naxis; ++k)
        points *= g.ncp[k];
    if (points > 7680)          // 60 KiB of LDS
        return hipErrorNotSupported;
    const size_t lds = (size_t)points * sizeof(double);
    int64_t nblk = (g.nvox + 255) / 256;
    if (nblk > 2048)
        nblk = 2048;
    const dim3 grid((unsigned)nblk);
    switch (g.naxis) {
    case 1: hipLaunchKernelGGL(source_box_kernel<1>, grid, dim3(256), lds, stream, g, box); break;
    case 2: hipLaunchKernelGGL(source_box_kernel<2>, grid, dim3(256), lds, stream, g, box); break;
    case 3: hipLaunchKernelGGL(source_box_kernel<3>, grid, dim3(256), lds, stream, g, box); break;
    case 4: hipLaunchKernelGGL(source_box_kernel<4>, grid, dim3(256), lds, stream, g, box); break;
    default: return hipErrorNotSupported;       // > 4 axes: callers filter the whole array
    }
    return hipGetLastError();
}

}  // namespace ed
