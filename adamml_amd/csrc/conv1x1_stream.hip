// Barrier-free streaming form of the algebraic data gradient of a layer-1 bottleneck conv3 / downsample conv
// (adamml_conv_bwd_data_alg with Cout = 256, Cin = 64):  dx[p, 0..64) = W_g [64 x 320] . [g'[p, 0..256) | a[p, 0..64)] + c_g.
//
// The activation operand never touches LDS: the B fragment of v_mfma_f32_16x16x32_bf16 is "lane (pixel li, k-chunk lg) holds 8
// consecutive channels of one pixel", which in NHWC is one 16-byte global load, so it goes global -> registers -> MFMA.  The
// per-group weight matrix (64 x 320 bf16 = 40 KB) is staged in LDS once per workgroup and read as the A operand; every wave
// then streams its own 32-pixel tiles with a 5-deep register ring and NO workgroup barrier in the loop.  The loader -> LDS ->
// barrier -> MFMA pipeline of conv_gemm_kernel's CAT instance ran this layer at 3.6 TB/s.
#include "common.h"
#include "../../include/adamml_hip.h"


namespace {

constexpr int K1 = 256, C2 = 64, KT = K1 + C2, KP = KT + 8;      // KP: padded LDS row (bank spread for the 16-lane row reads)
constexpr int NKS = KT / 32;                                      // 10 K steps: 8 from g', 2 from a
constexpr int PD = 4;                                             // K steps of global loads in flight per wave, issued in PAIRS (below)
constexpr int NPG = 2;                                            // 16-pixel groups per wave tile (64 px spilled 119 VGPRs)
constexpr int TPX = NPG * 16;

struct S1P {
    const bf16_t* g;         // [groups*P][256]
    const bf16_t* a;         // [groups*P][64] raw; value = act(scale * raw + shift) when a_scale != nullptr
    const float* a_scale;
    const float* a_shift;
    int a_act, a_gs;
    const bf16_t* w;         // [groups][64][320]
    const float* cadd;       // [groups][64]
    bf16_t* dx;              // [groups*P][64]
    int accumulate;
    const bf16_t* bn_z;      // [groups*P][64] or null: BatchNorm-fused epilogue (mask + sums) as adamml_conv_bwd_data_bn
    const float* bn_vec;     // [groups][4][64]
    int bn_act;
    double* stats;           // [groups][SLOTS][128]
    long P;                  // pixels per group
};

__global__ __launch_bounds__(256, 3) void alg_stream_kernel(S1P p) {
    __shared__ __attribute__((aligned(16))) bf16_t sw[C2 * KP];
    __shared__ float s_vec[2 * C2];          // lazy transform of a: scale, shift
    __shared__ float s_bn[4 * C2];           // scale, shift, mean, invstd of the epilogue BatchNorm
    __shared__ float s_add[C2];
    __shared__ float s_sum[4][2 * C2];       // per-wave rows of the BatchNorm-backward sums (common.h: reproducible reductions)
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    p.g += (size_t)g * p.P * K1;
    p.a += (size_t)g * p.P * C2;
    p.dx += (size_t)g * p.P * C2;
    p.w += (size_t)g * C2 * KT;
    p.cadd += (size_t)g * C2;
    if (p.a_scale) { p.a_scale += (size_t)g * p.a_gs; p.a_shift += (size_t)g * p.a_gs; }
    if (p.bn_z) { p.bn_z += (size_t)g * p.P * C2; p.bn_vec += (size_t)g * 4 * C2; p.stats += (size_t)g * ADAMML_STAT_SLOTS * 2 * C2; }
    for (int i = tid; i < C2 * (KT / 8); i += 256) {                // weights: 16-byte chunks, padded rows
        const int row = i / (KT / 8), ch = i - row * (KT / 8);
        *reinterpret_cast<bf16x8*>(&sw[row * KP + ch * 8]) = *reinterpret_cast<const bf16x8*>(p.w + (size_t)row * KT + ch * 8);
    }
    if (tid < C2) {
        s_vec[tid] = p.a_scale ? p.a_scale[tid] : 1.f;
        s_vec[C2 + tid] = p.a_scale ? p.a_shift[tid] : 0.f;
        s_add[tid] = p.cadd[tid];
#pragma unroll
        for (int w = 0; w < 4; ++w) { s_sum[w][tid] = 0.f; s_sum[w][C2 + tid] = 0.f; }
    }
    if (p.bn_z) s_bn[tid] = p.bn_vec[tid];                          // 256 threads == 4 * C2 entries
    __syncthreads();

    // (wave-uniform clamp bounds pinned to scalar registers: as vector registers one of them was spilled to scratch and re-read per tile)
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.a_scale ? act_lo(p.a_act) : -INFINITY), ahi = uniform(p.a_scale ? act_hi(p.a_act) : INFINITY);
    float ssum[4][4], ssq[4][4];                                    // [cout tile][r]: channel ct*16 + lg*4 + r
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[ct][r] = ssq[ct][r] = 0.f;

    const long ntile = (p.P + TPX - 1) / TPX;
    for (long t = (long)blockIdx.x * 4 + wave; t < ntile; t += (long)gridDim.x * 4) {
        const long p0 = t * TPX;
        long prow[NPG];                                               // this lane's pixel in each 16-pixel group (clamped)
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) {
            const long px = p0 + pg * 16 + li;
            prow[pg] = px < p.P ? px : p.P - 1;
        }
        f32x4 acc[NPG][4];                                          // [pixel group][cout tile]
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ring[PD][NPG];
        auto issue = [&](int slot, int k) {                         // k: K step (compile-time after unrolling)
#pragma unroll
            for (int pg = 0; pg < NPG; ++pg) {
                if (k < K1 / 32) ring[slot][pg] = *reinterpret_cast<const bf16x8*>(p.g + prow[pg] * K1 + k * 32 + lg * 8);      // (not non-temporal: the two 64-byte halves of a line are fetched by consecutive K steps and must meet in L2)
                else ring[slot][pg] = *reinterpret_cast<const bf16x8*>(p.a + prow[pg] * C2 + (k - K1 / 32) * 32 + lg * 8);
            }
        };
#pragma unroll
        for (int k = 0; k < PD; ++k) issue(k, k);
        // the epilogue's z rows (BatchNorm-fused form) / the rows accumulated into are requested HERE, behind the first K steps:
        // issued inside the epilogue each of the eight loads sat between two stores to dx, which the compiler may not reorder
        // (aliasing) and in-order vmcnt then exposes as eight dependent HBM round trips per tile
        bf16x4 zpre[NPG][4];
        if (p.bn_z || p.accumulate) {
            const bf16_t* src = p.bn_z ? p.bn_z : p.dx;
#pragma unroll
            for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) zpre[pg][ct] = *reinterpret_cast<const bf16x4*>(src + prow[pg] * C2 + ct * 16 + lg * 4);
        }
#pragma unroll
        for (int k = 0; k < NKS; ++k) {
            bf16x8 fb[NPG];
#pragma unroll
            for (int pg = 0; pg < NPG; ++pg) fb[pg] = ring[k % PD][pg];
            // the ring is refilled two K steps at a time: steps 2s and 2s + 1 are the two 64-byte halves of one 128-byte line of every pixel
            // (g' rows are four lines, a rows one), and requested back to back they cost the vector L1 one miss where requests a K step
            // apart cost two (profiles/r04_pmc_alg_stream.txt: 1.7 L2 requests per line, the L1 stalled 57 % of the time): 1.58 -> 1.55 ms
            // with a ring of 4 against the one-step refill of a ring of 5 (a paired ring of 6 does not fit 168 registers)
            if ((k & 1) && k - 1 + PD < NKS) { issue((k - 1) % PD, k - 1 + PD); issue(k % PD, k + PD); }
            if (k >= K1 / 32 && p.a_scale) {                        // lazy transform of the conv input: act(scale a + shift)
                const int c0 = (k - K1 / 32) * 32 + lg * 8;
                const f32x8 sc = load_f32x8(s_vec + c0), sh = load_f32x8(s_vec + C2 + c0);
#pragma unroll
                for (int pg = 0; pg < NPG; ++pg) {
                    f32x8 v = bf8_to_f32(fb[pg]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]), alo, ahi);
                    fb[pg] = f32_to_bf8(v);
                }
            }
            bf16x8 fa[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fa[ct] = *reinterpret_cast<const bf16x8*>(&sw[(ct * 16 + li) * KP + k * 32 + lg * 8]);
#pragma unroll
            for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[pg], acc[pg][ct], 0, 0, 0);
            // keep the unrolled K steps apart: without it the scheduler hoists every LDS fragment read and global load of the tile
            // to the top (256 VGPRs + scratch spills) and the register ring stops being a ring
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: lane (li, lg) holds channels ct*16 + lg*4 .. +3 of pixel pg*16 + li
        const float blo = uniform(act_lo(p.bn_act)), bhi = uniform(act_hi(p.bn_act));
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) {
            const long px = p0 + pg * 16 + li;
            const bool ok = px < p.P;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int ch = ct * 16 + lg * 4;
                f32x4 f = acc[pg][ct];
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] += s_add[ch + r];
                bf16_t* dst = p.dx + prow[pg] * C2 + ch;
                if (p.bn_z) {
                    const f32x4 zv = bf4_to_f32(zpre[pg][ct]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] *= mask_act(fmaf(zv[r], s_bn[ch + r], s_bn[C2 + ch + r]), blo, bhi);
                    const bf16x4 v = f32_to_bf4(f);
                    if (ok) *reinterpret_cast<bf16x4*>(dst) = v;
                    const f32x4 q = bf4_to_f32(v);
                    const float keep = ok ? 1.f : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gq = q[r] * keep;
                        ssum[ct][r] += gq;
                        ssq[ct][r] += gq * (zv[r] - s_bn[2 * C2 + ch + r]) * s_bn[3 * C2 + ch + r];
                    }
                } else {
                    if (p.accumulate) {
                        const f32x4 d0 = bf4_to_f32(zpre[pg][ct]);
                        f = bf4_to_f32(f32_to_bf4(f));                  // the GEMM kernels round the tile before accumulating
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] += d0[r];
                    }
                    if (ok) *reinterpret_cast<bf16x4*>(dst) = f32_to_bf4(f);
                }
            }
        }
    }
    if (p.bn_z) {
        // fold the 16 lanes (li) that share a channel set; one value per channel and wave, kept in the wave's own row
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = ssum[ct][r], b = ssq[ct][r];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (li == 0) {
                    s_sum[wave][ct * 16 + lg * 4 + r] = a;
                    s_sum[wave][C2 + ct * 16 + lg * 4 + r] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * C2)            // wave rows folded in wave order, one exact add per channel and workgroup
            stat_publish(p.stats + tid, 2 * C2, blockIdx.x & (ADAMML_STAT_SLOTS - 1), ((s_sum[0][tid] + s_sum[1][tid]) + s_sum[2][tid]) + s_sum[3][tid]);
    }
}

}  // namespace

bool adamml_alg_stream_supported(int Cout, int Cin) { return Cout == K1 && Cin == C2; }

int adamml_alg_stream_launch(const adamml_conv_desc_t* d, const void* g, const void* a, const float* a_scale, const float* a_shift,
                             const void* w_alg, const float* epi_add, void* dx, int accumulate, const void* z_in, const float* bn_vec,
                             int act, double* sums, hipStream_t stream) {
    S1P p;
    p.g = (const bf16_t*)g; p.a = (const bf16_t*)a; p.a_scale = a_scale; p.a_shift = a_shift; p.a_act = d->act; p.a_gs = d->in_gstride;
    p.w = (const bf16_t*)w_alg; p.cadd = epi_add; p.dx = (bf16_t*)dx; p.accumulate = accumulate;
    p.bn_z = (const bf16_t*)z_in; p.bn_vec = bn_vec; p.bn_act = act; p.stats = sums;
    p.P = (long)d->N * d->H * d->W;
    if (p.P <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const long ntile = (p.P + TPX - 1) / TPX;
    long nblk = (ntile + 3) / 4;
    const long cap = 768 / groups > 0 ? 768 / groups : 1;           // ~3 persistent workgroups per CU over all groups
    if (nblk > cap) nblk = cap;
    hipLaunchKernelGGL(alg_stream_kernel, dim3((unsigned)nblk, groups), dim3(256), 0, stream, p);
    return adamml_check_launch("conv_bwd_data_alg (stream)");
}
