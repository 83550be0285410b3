// dev tool ("the fused x + y prefilter pass, or its measured loss"): what would a fused x + y pass
// of the order-3 prefilter cost on a 256^3 float32 volume?  The shipped passes are two launches of 25 us each
// (profiles/r04_bench_kernel_stats.txt: prefilter_tile_strided / _contig, 5.1 TB/s of algorithmic bytes).  A fused pass
// keeps whole x-lines of T + 2 H consecutive y-rows of one z-slice in LDS (H = 16: the warm-up that makes a cut in a
// y-line invisible in float32), filters them along x in place, then along y, and writes the T inner rows.  This
// program times exactly that skeleton -- persistent workgroups, next tile's loads in flight in registers, row-contiguous
// 16-byte loads and stores, LDS-only barriers -- with the two filters replaced by the recursions' arithmetic on the
// tile (COMPUTE=1: per sample two fused multiply-adds forward and two backward along x, then the same along y, from and to
// LDS) or by nothing (COMPUTE=0: the data movement alone, the floor of any fused kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fused_xy.hip -o /tmp/ubench_fused_xy && /tmp/ubench_fused_xy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 256;           // volume side
constexpr int H = 16;            // warm-up rows on each side of a y-range
constexpr int kBlock = 256;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int T, int COMPUTE>
__global__ __launch_bounds__(kBlock) void fused_xy(const float* __restrict__ in, float* __restrict__ out, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];       // [T + 2 H][N + 4] (pitch / 4 odd)
    constexpr int R = T + 2 * H, P = N + 4;
    constexpr int CH = N / 4;                          // 16-byte chunks per row
    constexpr int NPF = R * CH / kBlock;               // loads per thread per tile
    const int tid = threadIdx.x;
    float4 v[NPF];
    auto issue = [&](int t) {
        const int z = t / (N / T), y0 = (t % (N / T)) * T - H;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int idx = u * kBlock + tid, r = idx / CH, c = idx % CH;
            int y = y0 + r;
            y = y < 0 ? -y : (y >= N ? 2 * N - 2 - y : y);          // mirror
            v[u] = *reinterpret_cast<const float4*>(in + ((size_t)z * N + y) * N + c * 4);
        }
    };
    int t = blockIdx.x;
    if (t >= ntiles)
        return;
    issue(t);
    for (;;) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int idx = u * kBlock + tid, r = idx / CH, c = idx % CH;
            *reinterpret_cast<float4*>(tile + r * P + c * 4) = v[u];
        }
        const int next = t + gridDim.x;
        if (next < ntiles)
            issue(next);
        lds_barrier();
        if (COMPUTE) {
            // x: one (row, 32-sample block) item per thread and round, recursion forward and backward, in place
            const float zp = -0.26794919f;
            for (int item = tid; item < R * (N / 32); item += kBlock) {
                float* row = tile + (item / (N / 32)) * P + (item % (N / 32)) * 32;
                float o[32];
                float yc = 0.f;
#pragma unroll
                for (int k = 0; k < 32; k += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(row + k);
                    yc = fmaf(zp, yc, x.x); o[k] = yc; yc = fmaf(zp, yc, x.y); o[k + 1] = yc;
                    yc = fmaf(zp, yc, x.z); o[k + 2] = yc; yc = fmaf(zp, yc, x.w); o[k + 3] = yc;
                }
                float ya = 0.f;
#pragma unroll
                for (int k = 31; k >= 0; --k) { ya = fmaf(zp, ya, o[k]); o[k] = ya; }
#pragma unroll
                for (int k = 0; k < 32; k += 4)
                    *reinterpret_cast<float4*>(row + k) = make_float4(o[k], o[k + 1], o[k + 2], o[k + 3]);
            }
            lds_barrier();
            // y: lane <-> column, blocks of T / 2 rows (with the warm-up rows in front and behind), in place
            for (int item = tid; item < N * 2; item += kBlock) {
                float* col = tile + (item % N) + (item / N) * (T / 2) * P;
                float yc = 0.f;
                for (int k = 0; k < H; ++k)
                    yc = fmaf(zp, yc, col[k * P]);
                float o[T / 2];
#pragma unroll
                for (int k = 0; k < T / 2; ++k) { yc = fmaf(zp, yc, col[(H + k) * P]); o[k] = yc; }
                float ya = 0.f;
                for (int k = H - 1; k >= 0; --k)
                    ya = fmaf(zp, ya, col[(H + T / 2 + k) * P]);
#pragma unroll
                for (int k = T / 2 - 1; k >= 0; --k) { ya = fmaf(zp, ya, o[k]); o[k] = ya; }

#pragma unroll
                for (int k = 0; k < T / 2; ++k)
                    col[(H + k) * P] = o[k];
            }
            lds_barrier();
        }
        {
            const int z = t / (N / T), y0 = (t % (N / T)) * T;
            for (int idx = tid; idx < T * CH; idx += kBlock) {
                const int r = idx / CH, c = idx % CH;
                *reinterpret_cast<float4*>(out + ((size_t)z * N + y0 + r) * N + c * 4) =
                    *reinterpret_cast<const float4*>(tile + (H + r) * P + c * 4);
            }
        }
        if (next >= ntiles)
            break;
        lds_barrier();
        t = next;
    }
}

template <int T, int COMPUTE>
int run(const float* in, float* out, int wgs_per_cu)
{
    constexpr int R = T + 2 * H, P = N + 4;
    const size_t lds = (size_t)R * P * sizeof(float);
    auto k = fused_xy<T, COMPUTE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ntiles = N * (N / T);
    const int grid = 256 * wgs_per_cu < ntiles ? 256 * wgs_per_cu : ntiles;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, 0, in, out, ntiles);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    const int iters = 20;
    for (int w = 0; w < iters; ++w)
        hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, 0, in, out, ntiles);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters;
    const double bytes = (double)N * N * N * 4 * (1.0 + (double)R / T);
    printf("T %3d rows (+%d halo)  LDS %6.1f KiB  %d workgroup(s) per CU  compute %d : %7.1f us per pass   %.2f TB/s moved (%.2fx read)\n",
           T, 2 * H, lds / 1024.0, wgs_per_cu, COMPUTE, us, bytes / us / 1e6, (double)R / T);
    return 0;
}

int main()
{
    float *in, *out;
    const size_t n = (size_t)N * N * N;
    CK(hipMalloc(&in, n * 4));
    CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 0, n * 4));
    printf("fused x + y prefilter pass, skeleton only, 256^3 float32 (the two shipped passes: 25 + 25 us)\n");
    if (run<32, 0>(in, out, 2)) return 1;
    if (run<32, 1>(in, out, 2)) return 1;
    if (run<64, 0>(in, out, 1)) return 1;
    if (run<64, 1>(in, out, 1)) return 1;
    return 0;
}
