// Probe: device-side cost of N DEPENDENT tiny kernels issued (a) as stream launches, (b) as one captured hipGraph, on one stream
// behind a pre-filled queue.  hipcc --offload-arch=gfx950 -O2 tools/graph_gap_probe.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long cycles) { const long t0 = clock64(); while (clock64() - t0 < cycles) { } }
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 1300, n = 1 << 16;
    float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int blocks : {1, 256, 2048}) {
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000000000L / 100);      // ~10 ms at 2 GHz: the queue fills behind it
            CK(hipEventRecord(a, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, n);
            CK(hipEventRecord(b, s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("stream  blocks %4d: %.3f ms device for %d launches = %.2f us each (host issue %.2f ms)\n", blocks, ms, N, ms * 1e3 / N,
                                 std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(a, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("graph   blocks %4d: %.3f ms device for %d nodes    = %.2f us each (host issue %.2f ms)\n", blocks, ms, N, ms * 1e3 / N,
                                 std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
