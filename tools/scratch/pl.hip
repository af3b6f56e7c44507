#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    unsigned l = threadIdx.x;
    unsigned a = 100 + l, b = 200 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto r2 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + l] = r2[0]; o[192 + l] = r2[1];
    o[256 + l] = __builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xf, 0xf, false);
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 320 * 4);
    k<<<1, 64>>>(d);
    unsigned h[320]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"p32 r0", "p32 r1", "p16 r0", "p16 r1", "ror8"};
    for (int t = 0; t < 5; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; i += 4) printf(" %u", h[t * 64 + i]); printf("\n"); }
    return 0;
}
