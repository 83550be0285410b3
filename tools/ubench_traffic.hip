// dev tool: calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on MI355X for the access patterns of the
// tile kernels (16-byte lanes in 64-byte runs; 4-byte lanes in 32-byte runs).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };

// read: every 16-byte chunk of buf exactly once; lanes 4q..4q+3 read one 64-byte run, consecutive
// runs of a wave are `run_stride` floats apart (mimics staging rows of a box)
__global__ void rd16_runs(const float* buf, size_t nchunks, int run_stride_chunks, float* sink)
{
    size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0;
    for (; id < nchunks; id += (size_t)gridDim.x * blockDim.x) {
        // permute runs so that neighbouring lanes-of-4 hit rows far apart
        size_t run = id >> 2, q = id & 3;
        size_t nruns = nchunks >> 2;
        size_t prun = (run * (size_t)run_stride_chunks) % nruns;   // stride coprime with nruns
        const F4u v = *reinterpret_cast<const F4u*>(buf + (prun * 4 + q) * 4);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == -1.f) *sink = acc;
}
__global__ void rd16_stream(const float4* buf, size_t n, float* sink)
{
    size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0;
    for (; id < n; id += (size_t)gridDim.x * blockDim.x) { float4 v = buf[id]; acc += v.x + v.w; }
    if (acc == -1.f) *sink = acc;
}
__global__ void rd4_stream(const float* buf, size_t n, float* sink)
{
    size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0;
    for (; id < n; id += (size_t)gridDim.x * blockDim.x) acc += buf[id];
    if (acc == -1.f) *sink = acc;
}
// write: 4-byte lanes; groups of 8 lanes write one 32-byte run; runs of a wave land in 8 different
// 128-byte lines (mimics the 8-wide x rows of an output tile), every byte written exactly once
__global__ void wr4_runs32(float* buf, size_t n)
{
    size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; id < n; id += (size_t)gridDim.x * blockDim.x) {
        size_t run = id >> 3, e = id & 7;
        size_t nruns = n >> 3;
        size_t prun = (run * 1031) % nruns;
        buf[prun * 8 + e] = 1.0f;
    }
}
__global__ void wr4_stream(float* buf, size_t n)
{
    size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; id < n; id += (size_t)gridDim.x * blockDim.x) buf[id] = 1.0f;
}
// read: K2's dY pattern.  buf is [rows][256 floats]; a wave-instruction reads 4 rows x 16 floats.
// MAP 0: lane = 16 * row + x (round 1 / early round 2); MAP 1: the 16-lane groups hold even / odd x
// of rows (y, y + 2) -- the mapping hot_grad_kernel uses.  Every element is read exactly once.
template <int MAP>
__global__ void rd4_k2rows(const float* buf, size_t nrows, float* sink)
{
    const int l = threadIdx.x & 63;
    const int xx = MAP ? 2 * (l & 7) + ((l >> 4) & 1) : (l & 15);
    const int yy = MAP ? ((l >> 5) & 1) + 2 * ((l >> 3) & 1) : (l >> 4);
    size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t nitems = (nrows / 4) * 16;       // (row group, 16-float chunk)
    float acc = 0;
    for (size_t it = wave; it < nitems; it += nwaves) {
        const size_t rg = it / 16, ch = it % 16;
        acc += buf[(rg * 4 + yy) * 256 + ch * 16 + xx];
    }
    if (acc == -1.f) *sink = acc;
}
// read: K1z's staging (round 6).  buf is [rows][256 floats]; a wave-instruction is ONE global_load_lds of 16 bytes per
// lane: CPR lanes cover a run of CPR * 16 bytes of a row (64-byte runs: 16 rows per instruction, 192-byte runs: 5 rows,
// 60 lanes), consecutive runs of an instruction are one row (1 KiB) apart.  SHIFT: the second copy of the box, the same
// runs one element (4 bytes) further -- a 16-byte lane then straddles two 16-byte chunks and the run two 64-byte lines.
// Every run is requested once per copy; coverage per row: 256 floats (CPR 4), 240 of 256 (CPR 12).
template <int CPR, bool SHIFT>
__global__ void lds16_box(const float* buf, size_t nrows, float* sink)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 256];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int RW = 64 / CPR;
    const int lr = l / CPR, q = l - lr * CPR;
    constexpr int NCH = 256 / (CPR * 4);
    size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t nitems = (nrows / RW) * NCH;
    float* dst = lds + w * 512;
    for (size_t it = wave; it < nitems; it += nwaves) {
        const size_t rg = it / NCH, ch = it % NCH;
        const float* g = buf + (rg * RW + lr) * 256 + ch * (CPR * 4) + q * 4;
        if (lr < RW) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            if (SHIFT)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 1),
                                                 (__attribute__((address_space(3))) void*)(dst + 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == -1.f) *sink = lds[threadIdx.x];
}
int main()
{
    const size_t bytes = (size_t)1 << 30;       // 1 GiB: larger than the 256 MiB Infinity Cache
    float* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    float* sink; CK(hipMalloc(&sink, 4));
    const size_t nf = bytes / 4, n16 = bytes / 16;
    hipLaunchKernelGGL(rd4_stream, dim3(4096), dim3(256), 0, 0, buf, nf, sink);
    hipLaunchKernelGGL(rd16_stream, dim3(4096), dim3(256), 0, 0, (const float4*)buf, n16, sink);
    hipLaunchKernelGGL(rd16_runs, dim3(4096), dim3(256), 0, 0, buf, n16, 1031, sink);
    hipLaunchKernelGGL(rd4_k2rows<0>, dim3(4096), dim3(256), 0, 0, buf, nf / 256, sink);
    hipLaunchKernelGGL(rd4_k2rows<1>, dim3(4096), dim3(256), 0, 0, buf, nf / 256, sink);
    hipLaunchKernelGGL((lds16_box<4, false>), dim3(4096), dim3(256), 0, 0, buf, nf / 256 - 1, sink);
    hipLaunchKernelGGL((lds16_box<4, true>), dim3(4096), dim3(256), 0, 0, buf, nf / 256 - 1, sink);
    hipLaunchKernelGGL((lds16_box<12, false>), dim3(4096), dim3(256), 0, 0, buf, nf / 256 - 1, sink);
    hipLaunchKernelGGL((lds16_box<12, true>), dim3(4096), dim3(256), 0, 0, buf, nf / 256 - 1, sink);
    hipLaunchKernelGGL(wr4_stream, dim3(4096), dim3(256), 0, 0, buf, nf);
    hipLaunchKernelGGL(wr4_runs32, dim3(4096), dim3(256), 0, 0, buf, nf);
    CK(hipDeviceSynchronize());
    printf("each kernel touches %zu bytes exactly once\n", bytes);
    return 0;
}
