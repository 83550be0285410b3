// Micro-benchmark (dev tool, not part of the product): throughput of float atomics on MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_atomics.hip -o gpurun_out/ubench_atomics && ./...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SCOPE>  // 0 agent, 1 workgroup, 2 plain store (non atomic, for reference)
__global__ void k_global(float* buf, size_t n, int iters, int stride_rows)
{
    // every wave walks over 64-float rows; consecutive waves take consecutive rows (coalesced,
    // no contention inside an iteration; every element is hit `iters` times overall)
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t nrows = n / 64;
    for (int it = 0; it < iters; ++it) {
        const size_t row = (wave * stride_rows + it * 7919) % nrows;
        float* p = buf + row * 64 + lane;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else *p = 1.0f;
    }
}

template <int MODE>  // 0 ds_add_f32 (no return), 1 plain ds read-modify-write
__global__ void k_lds(float* out, int iters)
{
    __shared__ float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int idx = (lane + it * 67) & 8191;
        if (MODE == 0) __hip_atomic_fetch_add(&s[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else s[idx] += 1.0f;
    }
    __syncthreads();
    float acc = 0;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) acc += s[i];
    if (acc == -1.f) out[0] = acc;
}

int main()
{
    const size_t n = 64u << 20;  // 64M floats = 256 MB
    float* buf; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 8, threads = 256, iters = 256;
    const double lane_ops = (double)blocks * threads * iters;
    for (int scope = 0; scope < 3; ++scope) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            if (scope == 0) hipLaunchKernelGGL(k_global<0>, dim3(blocks), dim3(threads), 0, 0, buf, n, iters, 1);
            if (scope == 1) hipLaunchKernelGGL(k_global<1>, dim3(blocks), dim3(threads), 0, 0, buf, n, iters, 1);
            if (scope == 2) hipLaunchKernelGGL(k_global<2>, dim3(blocks), dim3(threads), 0, 0, buf, n, iters, 1);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("global scope=%d (0 agent,1 workgroup,2 store): %.3f ms  %.1f G lane-ops/s  %.2f G row(256B)-ops/s  %.1f GB/s\n",
                            scope, ms, lane_ops / ms / 1e6, lane_ops / 64 / ms / 1e6, lane_ops * 4 / ms / 1e6);
        }
    }
    float* out; CK(hipMalloc(&out, 4));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            const int it2 = 4096;
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(256 * 4), dim3(256), 0, 0, out, it2);
            else hipLaunchKernelGGL(k_lds<1>, dim3(256 * 4), dim3(256), 0, 0, out, it2);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double ops = 256.0 * 4 * 256 * it2;
            if (rep) printf("lds mode=%d (0 ds_add_f32, 1 rmw): %.3f ms  %.1f G lane-ops/s  (%.2f wave-instr/clk/CU @2.4GHz)\n",
                            mode, ms, ops / ms / 1e6, ops / 64 / (ms * 1e-3) / 2.4e9 / 256);
        }
    }
    return 0;
}
