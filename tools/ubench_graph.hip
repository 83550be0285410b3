// dev tool: host cost of N dependent small kernel launches, directly and as a hipGraph replay with
// every node's parameters updated (hipGraphExecKernelNodeSetParams) -- the VERDICT r2 item 7 question.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_graph.hip -o /tmp/ubg && /tmp/ubg
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { float* p; int n; char pad[400]; };   // by-value argument block like HotGeom / TileGeom

__global__ void k(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < b.n) b.p[i] += 1.f; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int N = 7, iters = 2000;
    float* buf[2];
    hipMalloc(&buf[0], 1 << 20); hipMalloc(&buf[1], 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    Big b{}; b.p = buf[0]; b.n = 4096;
    for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, b);
    hipStreamSynchronize(s);
    // direct launches: host time to enqueue, and wall time per sequence with the queue drained at the end
    double t0 = now();
    for (int it = 0; it < iters; ++it) { b.p = buf[it & 1]; for (int j = 0; j < N; ++j) hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, b); }
    double t1 = now(); hipStreamSynchronize(s); double t2 = now();
    printf("direct: %d launches: host %.2f us per sequence (%.2f per launch), wall %.2f us per sequence\n", N, (t1 - t0) / iters, (t1 - t0) / iters / N, (t2 - t0) / iters);
    // graph: capture once, then per replay update every node and launch
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int j = 0; j < N; ++j) hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, b);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
    std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(g, nodes.data(), &nn);
    for (int w = 0; w < 50; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = now();
    for (int it = 0; it < iters; ++it) hipGraphLaunch(ge, s);
    t1 = now(); hipStreamSynchronize(s); t2 = now();
    printf("graph replay (no update): host %.2f us, wall %.2f us per sequence (%zu nodes)\n", (t1 - t0) / iters, (t2 - t0) / iters, nn);
    double tu = 0;
    t0 = now();
    for (int it = 0; it < iters; ++it) {
        b.p = buf[it & 1];
        void* args[] = {&b};
        hipKernelNodeParams kp{}; kp.func = (void*)k; kp.gridDim = dim3(16); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
        double a = now();
        for (size_t j = 0; j < nn; ++j) hipGraphExecKernelNodeSetParams(ge, nodes[j], &kp);
        tu += now() - a;
        hipGraphLaunch(ge, s);
    }
    t1 = now(); hipStreamSynchronize(s); t2 = now();
    printf("graph replay + %zu SetParams: host %.2f us (updates %.2f), wall %.2f us per sequence\n", nn, (t1 - t0) / iters, tu / iters, (t2 - t0) / iters);
    // one launch for comparison
    t0 = now();
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, b);
    t1 = now(); hipStreamSynchronize(s); t2 = now();
    printf("single kernel: host %.2f us, wall %.2f us\n", (t1 - t0) / iters, (t2 - t0) / iters);
    return 0;
}
