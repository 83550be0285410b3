// dev tool: LDS atomic / RMW throughput on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int iters)
{
    __shared__ unsigned long long s64[4096];
    unsigned int* s32 = (unsigned int*)s64;
    float* sf = (float*)s64;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s64[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int idx = (lane + it * 67) & 4095;
        if (MODE == 0) __hip_atomic_fetch_add(&sf[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 1) __hip_atomic_fetch_add(&s32[idx], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) __hip_atomic_fetch_add(&s64[idx], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 3) { sf[idx] += 1.0f; }                       // free-running rmw
        else if (MODE == 4) { sf[idx] += 1.0f; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }  // serialized rmw
        else if (MODE == 5) __hip_atomic_fetch_max(&s32[idx], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 6) { float2* p = (float2*)&sf[(idx & ~1)]; float2 v = *p; v.x += 1.f; v.y += 2.f; *p = v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    }
    __syncthreads();
    float acc = 0;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) acc += sf[i];
    if (acc == -1.f) out[0] = acc;
}

int main()
{
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "rmw free", "rmw serialized", "ds_max_u32", "rmw b64 serialized"};
    for (int threads : {64, 256}) {
        for (int mode = 0; mode < 7; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                const int it2 = 2048; const int blocks = 256 * (threads == 64 ? 8 : 4);
                CK(hipEventRecord(a));
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, it2); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, it2); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, it2); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, it2); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, out, it2); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, out, it2); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(threads), 0, 0, out, it2); break;
                }
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                const double ops = (double)blocks * threads * it2;
                if (rep) printf("threads=%3d %-20s: %.3f ms  %8.1f G lane-ops/s  %.1f cycles per wave-instr per CU (@2.4GHz, %d waves/CU)\n",
                                threads, names[mode], ms, ops / ms / 1e6,
                                (ms * 1e-3 * 2.4e9) / (ops / 64 / 256), blocks * threads / 64 / 256);
            }
        }
    }
    return 0;
}
