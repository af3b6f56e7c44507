// Probe: does an HBM streaming kernel that moves 8 bytes per lane and instruction reach the rate of one that moves 16?  Same traffic (two
// tensors read, one written, 1 GiB each), a thread walks `steps` steps with U accesses of each tensor in flight per step (the shape of the
// depthwise walkers).  hipcc --offload-arch=gfx950 -O3 tools/access_width_probe.hip -o /tmp/awp && /tmp/awp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <typename V, int U>
__global__ __launch_bounds__(256, 2) void walk(const V* a, const V* b, V* y, int steps, size_t stride) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int s = 0; s < steps; ++s) {
        V ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { ra[u] = a[i + u * stride]; rb[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = ra[u] + rb[u];
        i += (size_t)U * stride;
    }
}

static void *A, *B, *Y;
static hipEvent_t e0, e1;
static const size_t BYTES = (size_t)1 << 30;

template <typename V, int U>
static void run(const char* name, int steps) {
    const size_t nelem = BYTES / sizeof(V);
    const size_t threads = nelem / ((size_t)U * steps);
    const int nblk = (int)(threads / 256);
    const size_t stride = (size_t)nblk * 256;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((walk<V, U>), dim3(nblk), dim3(256), 0, 0, (const V*)A, (const V*)B, (V*)Y, steps, stride);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((walk<V, U>), dim3(nblk), dim3(256), 0, 0, (const V*)A, (const V*)B, (V*)Y, steps, stride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("%-26s U=%d steps=%3d blocks=%7d  %.3f ms  %.0f GB/s\n", name, U, steps, nblk, ms, 3.0 * BYTES / ms / 1e6);
}

int main() {
    (void)hipMalloc(&A, BYTES); (void)hipMalloc(&B, BYTES); (void)hipMalloc(&Y, BYTES);
    (void)hipMemset(A, 1, BYTES); (void)hipMemset(B, 2, BYTES);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int steps : {2, 8, 32}) {
        run<u2, 4>("8 B per lane (dwordx2)", steps);
        run<u4, 2>("16 B per lane (dwordx4)", steps);
        run<u2, 8>("8 B per lane (dwordx2)", steps);
        run<u4, 4>("16 B per lane (dwordx4)", steps);
    }
    return 0;
}
