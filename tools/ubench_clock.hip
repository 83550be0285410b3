// dev tool: what does s_memtime count, and what clock does the chip hold under VALU load?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip -o tools/ubench_clock.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long ticks, unsigned long long* out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0)
        out[0] = __builtin_readcyclecounter() - t0;
}
template <int KIND>
__global__ __launch_bounds__(256) void valu(int iters, float* out, unsigned long long* tk)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const float m = 1.000001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // 8 independent v_fma_f32
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        } else if (KIND == 1) {   // 4 independent v_fma_f64
            d0 = fma(d0, 1.0000001, 0.5); d1 = fma(d1, 1.0000001, 0.5); d2 = fma(d2, 1.0000001, 0.5); d3 = fma(d3, 1.0000001, 0.5);
        } else {                  // 8 independent integer-ish ops (v_cndmask)
            a0 = a1 > c ? a0 : a2; a1 = a2 > c ? a1 : a3; a2 = a3 > c ? a2 : a4; a3 = a4 > c ? a3 : a5;
            a4 = a5 > c ? a4 : a6; a5 = a6 > c ? a5 : a7; a6 = a7 > c ? a6 : a0; a7 = a0 > c ? a7 : a1;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0) tk[blockIdx.x] = t1 - t0;
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 1 << 20);
    float* o; hipMalloc(&o, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (unsigned long long t : {1000000ull, 10000000ull}) {
        hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, t, d); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("spin %llu s_memtime ticks: %.1f us -> %.1f ticks/us\n", t, ms * 1e3, t / (ms * 1e3));
    }
    auto run = [&](const char* name, auto kern, int per_iter, int wpe) {
        const int iters = 20000, blocks = 256 * wpe;       // wpe workgroups of 4 waves per CU = wpe waves per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, o, d); hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> tk(blocks); hipMemcpy(tk.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : tk) mean += v; mean /= blocks;
        const double instr_per_simd = (double)iters * per_iter * wpe;
        printf("%-28s %d waves/SIMD: %.1f us; %.2f ns per wave-instr per SIMD; kernel = %.0f ticks -> %.3f ticks per instr per SIMD, %.0f ticks/us\n",
               name, wpe, ms * 1e3, ms * 1e6 / instr_per_simd, mean, mean / instr_per_simd, mean / (ms * 1e3));
    };
    for (int w : {1, 2, 4}) {
        run("v_fma_f32 x8", valu<0>, 8, w);
        run("v_fma_f64 x4", valu<1>, 4, w);
        run("v_cmp+v_cndmask x8 (16 instr)", valu<2>, 16, w);
    }
    return 0;
}
