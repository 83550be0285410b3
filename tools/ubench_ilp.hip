// dev tool: VALU issue rate versus instruction-level parallelism (independent dependency chains per
// wave) and waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench_ilp.hip -o tools/ubench_ilp.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND, int CH>
__global__ __launch_bounds__(64) void chain(int iters, float* out, unsigned long long* tk)
{
    float a[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; d[i] = a[i]; }
    const float m = 1.000001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (KIND == 0) a[j] = fmaf(a[j], m, c);
                else if (KIND == 1) d[j] = fma(d[j], 1.0000001, 0.5);
                else { a[j] = (float)(int)a[j] + c; }          // v_cvt_i32_f32 + v_cvt_f32_i32 + v_add: dependent "other" ops
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) tk[blockIdx.x] = t1 - t0;
}
int main()
{
    unsigned long long* d; (void)hipMalloc(&d, 1 << 20);
    float* o; (void)hipMalloc(&o, 64 * 16384 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int per_iter, int wps) {
        const int iters = 20000, blocks = 256 * 4 * wps;
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, iters, o, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> tk(blocks); (void)hipMemcpy(tk.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : tk) mean += v; mean /= blocks;
        const double n = (double)iters * per_iter * wps;
        printf("%-34s waves/SIMD %d: %.3f ns, %.2f ticks per wave-instr per SIMD (%.0f MHz)\n", name, wps, ms * 1e6 / n, mean / n, mean / (ms * 1e3));
    };
    for (int w : {1, 3, 4, 8}) {
        run("v_fma_f32, 1 chain", chain<0, 1>, 8, w);
        run("v_fma_f32, 2 chains", chain<0, 2>, 8, w);
        run("v_fma_f32, 4 chains", chain<0, 4>, 8, w);
        run("v_fma_f32, 8 chains", chain<0, 8>, 8, w);
        run("v_fma_f64, 1 chain", chain<1, 1>, 8, w);
        run("v_fma_f64, 2 chains", chain<1, 2>, 8, w);
        run("v_fma_f64, 4 chains", chain<1, 4>, 8, w);
        run("cvt/cvt/add (3 instr), 1 chain", chain<2, 1>, 24, w);
        run("cvt/cvt/add (3 instr), 4 chains", chain<2, 4>, 24, w);
    }
    return 0;
}
