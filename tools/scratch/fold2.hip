#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)(1 << i) * (lane < 16 ? 1.f : lane < 32 ? 100.f : lane < 48 ? 10000.f : 1000000.f) / 1.f;
    // only one value active per test: use i==3 and i==6 markers
    float u[8], wv[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[2 * i]), __builtin_bit_cast(unsigned, v[2 * i + 1]), false, false);
        u[i] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    }
    for (int i = 0; i < 8; ++i) out[lane * 12 + i] = u[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, u[2 * j]), __builtin_bit_cast(unsigned, u[2 * j + 1]), false, false);
        wv[j] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    }
    for (int j = 0; j < 4; ++j) out[lane * 12 + 8 + j] = wv[j];
}
int main() {
    float* d; (void)hipMalloc(&d, 64 * 12 * 4);
    k<<<1, 64>>>(d);
    float h[64 * 12]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 16) { printf("lane %2d: u:", l); for (int i = 0; i < 8; ++i) printf(" %.0f", h[l * 12 + i]); printf(" | w:"); for (int j = 0; j < 4; ++j) printf(" %.0f", h[l * 12 + 8 + j]); printf("\n"); }
    return 0;
}
