// Streaming form of adamml_conv_fwd_bn_add for the ResNet-50 layer-2 bottlenecks (conv3 + bn3 + residual add + ReLU, models/resnet.py:104-112;
// 128 -> 512 channels at 28 x 28), gfx950.
//
// The tile kernel of conv_gemm.hip serves these launches at 3.5 TB/s (2.67 GB each at the benchmark shape, three per step).  Here the
// barrier-free structure of res_prod_stream.hip: wave q of an eight-wave workgroup owns the 64-channel slice q of the 512 output channels
// for the workgroup's 32-pixel tiles -- its rows of the identity operand and of the block output are 128-byte runs -- with its 64 x 128
// weight slice (16 MFMA A fragments) in registers for the whole kernel.  Per tile: the x fragments straight from global memory (one
// 16-byte load per lane and fragment, lazy bn2 + ReLU applied in registers; every wave loads them itself: L1 / L2 hits), 32 MFMAs, the
// raw tile rounded to bf16 into the wave's private LDS area (the tile kernel's rounding point), the epilogue in (pixel, 8-channel
// chunk) lanes -- scale3 z + shift3 + value(identity), ReLU, 16-byte store -- and the 1-bit mask gathered through LDS into one 4-byte
// store per lane.  The loads of the workgroup's next tile are requested after the stores.  Same K order, rounding points and epilogue
// expression as the tile kernel: out and mask_out are bit-identical.
#include "common.h"
#include "../../include/adamml_hip.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_;

struct FSP {
    const bf16_t* x;         // [groups][P][K] raw conv2 output
    const float* in_scale;   // its lazy BatchNorm (group stride in_gs) or null
    const float* in_shift;
    const bf16_t* w;         // [C][K] forward pack
    const float* bn_vec;     // [groups][4][C]: scale, shift of this conv's BatchNorm
    const bf16_t* idn;       // [groups][P][C] identity operand or null
    const float* id_scale;   // lazy identity (group stride id_gs) or null
    const float* id_shift;
    bf16_t* out;             // [groups][P][C]
    uint8_t* mask_out;       // [groups][P][C / 8] or null
    int in_act, in_gs, id_gs, act, P;
};

constexpr int TPX = 32;
constexpr int ZROW = 64 * 2 + 8;

template <int K, int NQ, bool MASK>
__global__ __launch_bounds__(NQ * 64, 1) void conv1x1_fadd_stream_kernel(FSP p) {
    constexpr int C = NQ * 64, KS = K / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_vec = reinterpret_cast<float*>(smem);                  // [2][K]
    float* s_bn = s_vec + 2 * K;                                    // [4][C]: scale, shift, id scale, id shift
    char* s_stage = reinterpret_cast<char*>(s_bn + 4 * C);          // [NQ waves][32][ZROW] + [NQ waves][256] mask bytes
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    {
        const size_t pp = (size_t)g * p.P;
        p.x += pp * K;
        p.out += pp * C + q * 64;
        p.idn += pp * C + q * 64;
        if (MASK) p.mask_out += pp * (C / 8) + q * 8;
    }
    for (int i = tid; i < K; i += NQ * 64) {
        s_vec[i] = p.in_scale ? p.in_scale[(size_t)g * p.in_gs + i] : 1.f;
        s_vec[K + i] = p.in_scale ? p.in_shift[(size_t)g * p.in_gs + i] : 0.f;
    }
    for (int i = tid; i < C; i += NQ * 64) {
        s_bn[i] = p.bn_vec[(size_t)g * 4 * C + i];
        s_bn[C + i] = p.bn_vec[(size_t)g * 4 * C + C + i];
        s_bn[2 * C + i] = p.id_scale ? p.id_scale[(size_t)g * p.id_gs + i] : 1.f;
        s_bn[3 * C + i] = p.id_scale ? p.id_shift[(size_t)g * p.id_gs + i] : 0.f;
    }
    char* zs = s_stage + q * (TPX * ZROW);
    char* ms = s_stage + NQ * (TPX * ZROW) + q * 256;
    __syncthreads();
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.in_act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.in_act) : INFINITY);
    const float rlo = uniform(act_lo(p.act)), rhi = uniform(act_hi(p.act));
    const bool lazy = p.in_scale != nullptr;
    // ---- this wave's weight slice: A fragments (row = output channel 64 q + 16 ct + li, k = 32 ks + 8 lg ..)
    bf16x8 wr[4][KS];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wr[ct][ks] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(q * 64 + ct * 16 + li) * K + ks * 32 + lg * 8);
    const int ntile = (p.P + TPX - 1) / TPX;
    // this lane's four (pixel, 8-channel chunk) slots of a tile: chunk lane % 8 of pixels lane / 8 + 8 i
    const int zch = lane & 7, zpx = lane >> 3;
    const int c0 = q * 64 + zch * 8;
    const f32x8 sc = load_f32x8(s_bn + c0), sh = load_f32x8(s_bn + C + c0), isc = load_f32x8(s_bn + 2 * C + c0), ish = load_f32x8(s_bn + 3 * C + c0);

    bf16x8 rx[2][KS], ri[4];
    // (uniform 64-bit bases -- the tile is the same for the whole wave -- plus 32-bit lane offsets)
    auto issue = [&](int tile) {
        const int p0 = tile * TPX;
        const int npx = p.P - p0 < TPX ? p.P - p0 : TPX;
        const char* xb = reinterpret_cast<const char*>(p.x + (size_t)p0 * K);
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int px = pg * 16 + li, pc = px < npx ? px : npx - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) rx[pg][ks] = *reinterpret_cast<const bf16x8*>(xb + (unsigned)((pc * K + ks * 32 + lg * 8) * 2));
        }
        const char* ib = reinterpret_cast<const char*>(p.idn + (size_t)p0 * C);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = zpx + 8 * i, pc = px < npx ? px : npx - 1;
            ri[i] = *reinterpret_cast<const bf16x8*>(ib + (unsigned)((pc * C + zch * 8) * 2));
        }
    };
    issue((int)blockIdx.x < ntile ? (int)blockIdx.x : ntile - 1);
    // one tile; FULL (a compile-time flag: 32 pixels) keeps the stores unconditional -- a store under a per-lane condition makes every
    // wait behind it a conservative one
    auto body = [&](int tile, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int p0 = tile * TPX;
        const int npx = FULL ? TPX : p.P - p0;
        f32x4 c[2][4];
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) c[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x8 vs = load_f32x8(s_vec + ks * 32 + lg * 8), vh = load_f32x8(s_vec + K + ks * 32 + lg * 8);
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                bf16x8 fb = rx[pg][ks];
                if (lazy) {
                    f32x8 v = bf8_to_f32(fb);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], vs[i], vh[i]), alo, ahi);
                    fb = f32_to_bf8(v);
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) c[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[ct][ks], fb, c[pg][ct], 0, 0, 0);
            }
        }
        // raw tile as bf16 (the tile kernel's rounding point): lane (li, lg) holds channels 16 ct + 4 lg .. + 3 of pixel 16 pg + li
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                union { bf16x4 b; s16x4_ s; } u;
                u.b = f32_to_bf4(c[pg][ct]);
                *reinterpret_cast<s16x4_*>(zs + (pg * 16 + li) * ZROW + (ct * 16 + lg * 4) * 2) = u.s;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (own LDS writes landed; no other wave touches this area)
        char* ob = reinterpret_cast<char*>(p.out + (size_t)p0 * C);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = zpx + 8 * i;
            const char* zp = zs + px * ZROW + zch * 16;
            union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
            u.s.a = *reinterpret_cast<const s16x4_*>(zp);
            u.s.b = *reinterpret_cast<const s16x4_*>(zp + 8);
            f32x8 f = bf8_to_f32(u.v);
            const f32x8 w = bf8_to_f32(ri[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = clamp_act(fmaf(f[j], sc[j], sh[j]) + fmaf(w[j], isc[j], ish[j]), rlo, rhi);
            const bf16x8 v = f32_to_bf8(f);
            if (FULL || px < npx) *reinterpret_cast<bf16x8*>(ob + (unsigned)((px * C + zch * 8) * 2)) = v;
            if (MASK) {
                const f32x8 qv = bf8_to_f32(v);
                unsigned bits = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) bits |= (qv[j] > rlo && qv[j] < rhi) ? (1u << j) : 0u;
                reinterpret_cast<uint8_t*>(ms)[px * 8 + zch] = (uint8_t)bits;
            }
        }
        if (MASK) {
            // the wave's 32 x 8 mask bytes: one 4-byte store per lane (pixel lane / 2, half lane % 2 of its 8-byte run)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned m = reinterpret_cast<const unsigned*>(ms)[lane];
            const int px = lane >> 1;
            if (FULL || px < npx) *reinterpret_cast<unsigned*>(p.mask_out + (size_t)(p0 + px) * (C / 8) + (lane & 1) * 4) = m;
        }
        // (unconditional request of the workgroup's next tile; past the end: this tile again, unused)
        issue(tile + (int)gridDim.x < ntile ? tile + (int)gridDim.x : tile);
    };
    {
        const int nfull = p.P / TPX;                                         // (at most one partial tile: the last)
        int tile = blockIdx.x;
#pragma unroll 1
        for (; tile < nfull; tile += gridDim.x) body(tile, std::true_type{});
        if (tile < ntile) body(tile, std::false_type{});
    }
}

// The same conv + BatchNorm + add + ReLU with the temporal max-pool behind the last block of a stage in the epilogue
// (adamml_conv_fwd_bn_add_tpool: kernel 3 / stride 2 / pad 1 over the T frames of a clip).  A task = (clip, 16-pixel block); the 32 rows of a
// step are the block's pixels in the two frames (2 to, 2 to + 1) of window `to` -- rows 0..15 the even frame's, 16..31 the odd frame's, the
// layout of tpool_bwd_prod.hip -- so one step of the (run-time) window loop is a whole window: tap 0 = the odd frame of the previous step,
// carried in registers (-inf before window 0: every ReLU output beats it), tap 1 / tap 2 = this step's two frames; first maximum in scan
// order, code 3 where the maximum does not pass the ReLU, exactly the TP epilogue of conv_gemm_kernel: pooled and code are bit-identical,
// the full-rate block output never exists.  Every step stores (no store under a run-time condition); the 16-bit codes are gathered
// through LDS into one 4-byte store per lane.  The next step's rows (this task's next window, or window 0 of the workgroup's next task)
// are requested after the stores.
struct FTS {
    const bf16_t* x; const float* in_scale; const float* in_shift; const bf16_t* w; const float* bn_vec;
    const bf16_t* idn; const float* id_scale; const float* id_shift;
    bf16_t* pooled;          // [groups][clips * T / 2][HW][C]
    uint16_t* code;          // [groups][clips * T / 2][HW][C / 8] or null
    int in_act, in_gs, id_gs, act, T, HW, clips;
};

template <int K, int NQ, bool CODE>
__global__ __launch_bounds__(NQ * 64, NQ == 4 ? 2 : 1) void conv1x1_fadd_tpool_stream_kernel(FTS p) {
    constexpr int C = NQ * 64, KS = K / 32, FPX = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_vec = reinterpret_cast<float*>(smem);                  // [2][K]
    float* s_bn = s_vec + 2 * K;                                    // [4][C]: scale, shift, id scale, id shift
    char* s_stage = reinterpret_cast<char*>(s_bn + 4 * C);          // [NQ waves][32][ZROW] + [NQ waves][256] code bytes
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int To = p.T >> 1;
    {
        const size_t P = (size_t)p.clips * p.T * p.HW, Pp = (size_t)p.clips * To * p.HW;
        p.x += (size_t)g * P * K;
        p.idn += (size_t)g * P * C + q * 64;
        p.pooled += (size_t)g * Pp * C + q * 64;
        if (CODE) p.code += (size_t)g * Pp * (C / 8) + q * 8;
    }
    for (int i = tid; i < K; i += NQ * 64) {
        s_vec[i] = p.in_scale ? p.in_scale[(size_t)g * p.in_gs + i] : 1.f;
        s_vec[K + i] = p.in_scale ? p.in_shift[(size_t)g * p.in_gs + i] : 0.f;
    }
    for (int i = tid; i < C; i += NQ * 64) {
        s_bn[i] = p.bn_vec[(size_t)g * 4 * C + i];
        s_bn[C + i] = p.bn_vec[(size_t)g * 4 * C + C + i];
        s_bn[2 * C + i] = p.id_scale ? p.id_scale[(size_t)g * p.id_gs + i] : 1.f;
        s_bn[3 * C + i] = p.id_scale ? p.id_shift[(size_t)g * p.id_gs + i] : 0.f;
    }
    char* zs = s_stage + q * (TPX * ZROW);
    char* ms = s_stage + NQ * (TPX * ZROW) + q * 256;
    __syncthreads();
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.in_act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.in_act) : INFINITY);
    const float rlo = uniform(act_lo(p.act)), rhi = uniform(act_hi(p.act));
    const bool lazy = p.in_scale != nullptr;
    bf16x8 wr[4][KS];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wr[ct][ks] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(q * 64 + ct * 16 + li) * K + ks * 32 + lg * 8);
    const int nblk_px = (p.HW + FPX - 1) / FPX;                      // 16-pixel blocks per frame
    const int ntask = p.clips * nblk_px;
    // this lane's slots: chunk lane % 8 of pixels lane / 8 and lane / 8 + 8, in the even frame (i = 0, 1) and the odd frame (i = 2, 3)
    const int zch = lane & 7, zpx = lane >> 3;
    const int c0 = q * 64 + zch * 8;
    const f32x8 sc = load_f32x8(s_bn + c0), sh = load_f32x8(s_bn + C + c0), isc = load_f32x8(s_bn + 2 * C + c0), ish = load_f32x8(s_bn + 3 * C + c0);

    bf16x8 rx[2][KS], ri[4];
    // window `to` of task `task`: (uniform 64-bit bases per frame + 32-bit lane offsets; pixels past the end of a frame's last block clamped)
    auto issue = [&](int task, int to) {
        const int clip = task / nblk_px, blk = task - clip * nblk_px;
        const int npx = p.HW - blk * FPX < FPX ? p.HW - blk * FPX : FPX;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const size_t base = ((size_t)clip * p.T + 2 * to + f) * p.HW + (size_t)blk * FPX;
            const char* xb = reinterpret_cast<const char*>(p.x + base * K);
            const int pc = li < npx ? li : npx - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) rx[f][ks] = *reinterpret_cast<const bf16x8*>(xb + (unsigned)((pc * K + ks * 32 + lg * 8) * 2));
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const size_t base = ((size_t)clip * p.T + 2 * to + f) * p.HW + (size_t)blk * FPX;
            const char* ib = reinterpret_cast<const char*>(p.idn + base * C);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int px = zpx + 8 * s2, pc = px < npx ? px : npx - 1;
                ri[2 * f + s2] = *reinterpret_cast<const bf16x8*>(ib + (unsigned)((pc * C + zch * 8) * 2));
            }
        }
    };
    issue((int)blockIdx.x < ntask ? (int)blockIdx.x : ntask - 1, 0);
    auto body = [&](int task, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int clip = task / nblk_px, blk = task - clip * nblk_px;
        const int npx = FULL ? FPX : p.HW - blk * FPX;
        bf16x8 carry[2];                                                    // tap 0 of the open window: the previous step's odd frame
        {
            const __bf16 ninf = (__bf16)(-INFINITY);
            carry[0] = carry[1] = bf16x8{ninf, ninf, ninf, ninf, ninf, ninf, ninf, ninf};
        }
#pragma unroll 1
        for (int to = 0; to < To; ++to) {
            // (one frame = 16-pixel group at a time: 16 accumulator registers live instead of 32)
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                f32x4 c[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) c[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 fb = rx[pg][ks];
                    if (lazy) {
                        const f32x8 vs = load_f32x8(s_vec + ks * 32 + lg * 8), vh = load_f32x8(s_vec + K + ks * 32 + lg * 8);
                        f32x8 v = bf8_to_f32(fb);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], vs[i], vh[i]), alo, ahi);
                        fb = f32_to_bf8(v);
                    }
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) c[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[ct][ks], fb, c[ct], 0, 0, 0);
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    union { bf16x4 b; s16x4_ s; } u;
                    u.b = f32_to_bf4(c[ct]);
                    *reinterpret_cast<s16x4_*>(zs + (pg * 16 + li) * ZROW + (ct * 16 + lg * 4) * 2) = u.s;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (own LDS writes landed; no other wave touches this area)
            bf16x8 vb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (i >> 1) * 16 + zpx + 8 * (i & 1);
                const char* zp = zs + row * ZROW + zch * 16;
                union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
                u.s.a = *reinterpret_cast<const s16x4_*>(zp);
                u.s.b = *reinterpret_cast<const s16x4_*>(zp + 8);
                f32x8 f = bf8_to_f32(u.v);
                const f32x8 w = bf8_to_f32(ri[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = clamp_act(fmaf(f[j], sc[j], sh[j]) + fmaf(w[j], isc[j], ish[j]), rlo, rhi);
                vb[i] = f32_to_bf8(f);                                      // the value the unfused path stores and the pool re-reads
            }
            char* ob = reinterpret_cast<char*>(p.pooled + (((size_t)clip * To + to) * p.HW + (size_t)blk * FPX) * C);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int px = zpx + 8 * s2;
                f32x8 bst = bf8_to_f32(carry[s2]);
                const f32x8 v1 = bf8_to_f32(vb[s2]), v2 = bf8_to_f32(vb[2 + s2]);
                unsigned cd = 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {                               // first maximum in scan order: taps 0, 1, 2
                    if (v1[j] > bst[j]) { bst[j] = v1[j]; cd = (cd & ~(3u << (2 * j))) | (1u << (2 * j)); }
                    if (v2[j] > bst[j]) { bst[j] = v2[j]; cd = (cd & ~(3u << (2 * j))) | (2u << (2 * j)); }
                    if (!(bst[j] > rlo && bst[j] < rhi)) cd |= 3u << (2 * j);                   // act'(maximum) == 0: no gradient through this window
                }
                if (FULL || px < npx) *reinterpret_cast<bf16x8*>(ob + (unsigned)((px * C + zch * 8) * 2)) = f32_to_bf8(bst);
                if (CODE) reinterpret_cast<uint16_t*>(ms)[px * 8 + zch] = (uint16_t)cd;
                carry[s2] = vb[2 + s2];                                     // tap 0 of the next window
            }
            if constexpr (CODE) {
                // the wave's 16 x 8 code words: one 4-byte store per lane (pixel lane / 4, quarter lane % 4 of its 16-byte run)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned m = reinterpret_cast<const unsigned*>(ms)[lane];
                const int px = lane >> 2;
                if (FULL || px < npx)
                    *reinterpret_cast<unsigned*>(p.code + (((size_t)clip * To + to) * p.HW + (size_t)blk * FPX + px) * (C / 8) + (lane & 3) * 2) = m;
            }
            // (unconditional request of the next step: this task's next window, or window 0 of the workgroup's next task; past the end: again)
            {
                int tn = task, wn = to + 1;
                if (wn == To) { wn = 0; tn = task + (int)gridDim.x; if (tn >= ntask) { tn = task; wn = To - 1; } }
                issue(tn, wn);
            }
        }
    };
#pragma unroll 1
    for (int task = blockIdx.x; task < ntask; task += gridDim.x) {
        const int blk = task % nblk_px;
        if (p.HW - blk * FPX >= FPX) body(task, std::true_type{});
        else body(task, std::false_type{});
    }
}

int fs_blocks(long P, int groups) {
    const long ntile = (P + TPX - 1) / TPX;
    static const long cap0 = getenv("ADAMML_FADD_STREAM_CAP") ? atol(getenv("ADAMML_FADD_STREAM_CAP")) : 256;    // A/B aid
    long cap = cap0 / (groups < 1 ? 1 : groups);
    if (cap < 1) cap = 1;
    return (int)(ntile < cap ? ntile : cap);
}

bool fs_on() { const char* e = getenv("ADAMML_FADD_STREAM"); return !(e && atoi(e) == 0); }                      // A/B aid, read at every call

}  // namespace

// (declared in conv_gemm.hip, which owns the C entry point and falls back to its tile kernel)
int adamml_conv1x1_fadd_stream_supported(const adamml_conv_desc_t* d) {
    if (!fs_on() || !d) return 0;
    return d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->up <= 1 && d->Cin == 128 && d->Cout == 512 &&
           (long)d->N * d->OH * d->OW >= 4096 ? 1 : 0;                       // (and an identity operand: the launcher's caller checks)
}

int adamml_conv1x1_fadd_stream_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                      const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                      void* out, uint8_t* mask_out, hipStream_t stream) {
    constexpr int K = 128, NQ = 8, C = NQ * 64;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const long P = (long)d->N * d->OH * d->OW;
    FSP p;
    p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr; p.w = (const bf16_t*)w_packed; p.bn_vec = bn_vec;
    p.idn = (const bf16_t*)idn; p.id_scale = idn ? id_scale : nullptr; p.id_shift = idn && id_scale ? id_shift : nullptr;
    p.out = (bf16_t*)out; p.mask_out = mask_out;
    p.in_act = d->act; p.in_gs = d->in_gstride; p.id_gs = id_gstride; p.act = act; p.P = (int)P;
    constexpr size_t lds = (2 * K + 4 * C) * 4 + (size_t)NQ * (TPX * ZROW + 256);
    const dim3 grid((unsigned)fs_blocks(P, groups), groups);
    if (mask_out) hipLaunchKernelGGL((conv1x1_fadd_stream_kernel<K, NQ, true>), grid, dim3(NQ * 64), lds, stream, p);
    else hipLaunchKernelGGL((conv1x1_fadd_stream_kernel<K, NQ, false>), grid, dim3(NQ * 64), lds, stream, p);
    return adamml_check_launch("conv_fwd_bn_add(stream)");
}

// d: the forward descriptor of conv3 (N = clips * frames images per group)
int adamml_conv1x1_fadd_tpool_stream_supported(const adamml_conv_desc_t* d, int frames) {
    const char* e = getenv("ADAMML_FADD_TPOOL_SLICE");                                                             // A/B aid, read at every call
    const int mode = e ? atoi(e) : 1;                                                                              // 0: off, 1: layer 2, 2: layers 1 and 2
    if (!mode || !d) return 0;
    const bool l2 = d->Cin == 128 && d->Cout == 512, l1 = mode >= 2 && d->Cin == 64 && d->Cout == 256;
    return d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->up <= 1 && (l1 || l2) && (frames == 2 || frames == 4 || frames == 8) &&
           d->N % frames == 0 && (long)d->N * d->OH * d->OW >= 4096 ? 1 : 0;
}

int adamml_conv1x1_fadd_tpool_stream_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                            const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                            int frames, void* pooled, uint16_t* code, hipStream_t stream) {
    const int groups = d->groups < 1 ? 1 : d->groups;
    FTS p;
    p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr; p.w = (const bf16_t*)w_packed; p.bn_vec = bn_vec;
    p.idn = (const bf16_t*)idn; p.id_scale = id_scale; p.id_shift = id_scale ? id_shift : nullptr; p.pooled = (bf16_t*)pooled; p.code = code;
    p.in_act = d->act; p.in_gs = d->in_gstride; p.id_gs = id_gstride; p.act = act; p.T = frames; p.HW = d->OH * d->OW; p.clips = d->N / frames;
    const long ntask = (long)p.clips * ((p.HW + 15) / 16);
    const bool l2 = d->Cin == 128;
    long cap = (l2 ? 256 : 512) / groups;
    if (cap < 1) cap = 1;
    const dim3 grid((unsigned)(ntask < cap ? ntask : cap), groups);
    if (l2) {
        constexpr int K = 128, NQ = 8, C = NQ * 64;
        constexpr size_t lds = (2 * K + 4 * C) * 4 + (size_t)NQ * (TPX * ZROW + 256);
        if (code) hipLaunchKernelGGL((conv1x1_fadd_tpool_stream_kernel<K, NQ, true>), grid, dim3(NQ * 64), lds, stream, p);
        else hipLaunchKernelGGL((conv1x1_fadd_tpool_stream_kernel<K, NQ, false>), grid, dim3(NQ * 64), lds, stream, p);
    } else {
        constexpr int K = 64, NQ = 4, C = NQ * 64;
        constexpr size_t lds = (2 * K + 4 * C) * 4 + (size_t)NQ * (TPX * ZROW + 256);
        if (code) hipLaunchKernelGGL((conv1x1_fadd_tpool_stream_kernel<K, NQ, true>), grid, dim3(NQ * 64), lds, stream, p);
        else hipLaunchKernelGGL((conv1x1_fadd_tpool_stream_kernel<K, NQ, false>), grid, dim3(NQ * 64), lds, stream, p);
    }
    return adamml_check_launch("conv_fwd_bn_add_tpool(stream)");
}
