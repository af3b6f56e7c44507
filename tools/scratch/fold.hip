#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CPR>
__global__ void k(const float* in, float* cs) {   // in[lane][16], cs[2][CPR*8]
    const int lane = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[lane * 16 + i];
    if (CPR == 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x128, 0xf, 0xf, false));
    }
    float u[8], wv[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        u[i] = a + b;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = u[2 * j], b = u[2 * j + 1];
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        wv[j] = a + b;
    }
    const int lrow = lane >> 4, vsel = ((lrow & 1) << 1) | (lrow >> 1), ech = lane % CPR;
    if (CPR >= 16 || !(lane & 8)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int id = 4 * j + vsel;
            atomicAdd(&cs[(id >> 3) * CPR * 8 + ech * 8 + (id & 7)], wv[j]);
        }
    }
}
template <int CPR> int run() {
    float h[64 * 16], ref[2 * 128] = {0}, out[2 * 128];
    for (int i = 0; i < 64 * 16; ++i) h[i] = (float)((i * 7919) % 101) - 50.f;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) ref[(i >> 3) * CPR * 8 + (l % CPR) * 8 + (i & 7)] += h[l * 16 + i];
    float *din, *dcs; (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dcs, sizeof(out));
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice); (void)hipMemset(dcs, 0, sizeof(out));
    k<CPR><<<1, 64>>>(din, dcs);
    (void)hipMemcpy(out, dcs, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2 * CPR * 8; ++i) if (out[i] != ref[i]) { if (bad < 5) printf("CPR=%d i=%d got %f want %f\n", CPR, i, out[i], ref[i]); ++bad; }
    printf("CPR=%d bad=%d\n", CPR, bad);
    return bad;
}
int main() { return run<16>() + run<8>(); }
